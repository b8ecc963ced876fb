#!/usr/bin/env python3
"""Rewrites the results table of DESIGN.md section 5 from profiles/r02_bench_final.json (development tool)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_final.json")))
e = d["extra"]
t = lambda k: e[k]["tokens_per_s"]
f = lambda k: 100 * e[k].get("frac_of_hbm_peak", e[k].get("frac_of_mfma_peak"))
c = lambda v: "{:,.0f}".format(v)
rows = []
rows.append("| fp16 decode B=1 ctx 2048 (headline, configs[2]) | **%.0f tok/s**, %.2f ms/step (358-366 on different boxes) | 14.29 GB/step -> %.1f TB/s = %.0f %% of HBM peak (a 1 GiB device copy on the same box: %.1f TB/s); gate/up GEMV %.0f %% by hipEvents, %.0f %% by kernel trace (%.1f us) | 362-369 tok/s, 65-66 %% |" % (d["value"], d["ms_per_step"], d["whole_step"]["achieved_GBs"] / 1e3, 100 * d["whole_step"]["frac_of_hbm_peak"], d["peaks"]["measured_copy_GBs"] / 1e3, 100 * d["roofline"]["frac_event"], 100 * d["roofline"]["frac_rocprof_trace"], d["roofline"]["rocprof_avg_us"]))
rows.append("| fp16 decode B=1 ctx 128 | %.0f tok/s | %.0f %% | 390, 65 %% |" % (t("decode_f16_b1_ctx128"), f("decode_f16_b1_ctx128")))
rows.append("| int8 W-only decode B=1 ctx 2048 | %.0f tok/s | %.0f %% (7.81 GB/step) | 500, 49 %% |" % (t("decode_int8_b1_ctx2048"), f("decode_int8_b1_ctx2048")))
ri = d["roofline_int8"]
rows.append("| int8 W-only decode B=32 ctx 128 (configs[3]) | **%s tok/s** | %.0f %% (8.90 GB/step); gate/up packed kernel %.0f %% by events, %.0f %% by trace, PMC traffic %.2fx the weight bytes | 10,080, 35 %% |" % (c(t("decode_int8_b32_ctx128")), f("decode_int8_b32_ctx128"), 100 * ri["frac_event"], 100 * ri["frac_rocprof_trace"], ri["traffic"] / ri["algorithmic_work_per_launch"]))
rows.append("| int4 (group 128) decode B=1 ctx 2048 | **%.0f tok/s** (int8: %.0f) | %.0f %% of its 4.68 GB/step | 467, 27 %% |" % (t("decode_int4_b1_ctx2048"), t("decode_int8_b1_ctx2048"), f("decode_int4_b1_ctx2048")))
rows.append("| int4 decode B=32 ctx 128 | **%s tok/s** (int8: %s) | %.0f %% of its 5.77 GB/step (same launches and fixed costs as int8 at half the weight bytes) | 9.0k, 20 %% |" % (c(t("decode_int4_b32_ctx128")), c(t("decode_int8_b32_ctx128")), f("decode_int4_b32_ctx128")))
bs = [1, 2, 4, 8, 16, 32, 64, 128]
rows.append("| fp8 decode B=1/2/4/8/16/32/64/128 ctx 512 (configs[4]) | %s tok/s | %s %% | 515 / 815 / 1,388 / 2,516 / 4,374 / 7,058 / 10,510 / 13,240-14,000 |" % (" / ".join(c(t("decode_fp8_b%d_ctx512" % b)) for b in bs), " / ".join("%.0f" % f("decode_fp8_b%d_ctx512" % b) for b in bs)))
rows.append("| fp16 decode B=32/128 ctx 512 | %s tok/s | %.0f / %.0f %% | 6,240 / 12,410-12,810 |" % (" / ".join(c(t(k)) for k in ("decode_f16_b32_ctx512", "decode_f16_b128_ctx512")), f("decode_f16_b32_ctx512"), f("decode_f16_b128_ctx512")))
rows.append("| int8 decode B=32 ctx 2048, fp16 KV / e4m3 KV | %s tok/s | %.0f %% (41.1 GB/step) / **%.0f %%** (23.9 GB/step) | 4,005 / 4,858, 64 / 45 %% |" % (" / ".join(c(t(k)) for k in ("decode_int8_b32_ctx2048", "decode_int8_b32_ctx2048_kvfp8")), f("decode_int8_b32_ctx2048"), f("decode_int8_b32_ctx2048_kvfp8")))
rows.append("| fp16 decode B=1 ctx 2048 with the e4m3 KV cache | %.0f tok/s | %.0f %% (13.75 GB/step) | 369-371 |" % (t("decode_f16_b1_ctx2048_kvfp8"), f("decode_f16_b1_ctx2048_kvfp8")))
rows.append("| fp8 weights + e4m3 KV cache, B=128 ctx 512 | %s tok/s (fp16 KV: %s) | %.0f %% (23.9 GB/step) | 15,840-15,970 |" % (c(t("decode_fp8_b128_ctx512_kvfp8")), c(t("decode_fp8_b128_ctx512")), f("decode_fp8_b128_ctx512_kvfp8")))
p = d["peaks"]
rows.append("| fp16 prefill b1 s2048 / b8 s512 / b1 s128 (configs[1]: the 128-token case) | **%s tok/s** | %.0f / %.0f / %.0f %% of 2.5 PFLOP/s; gate/up GEMM %.0f %% (MFMA busy 68 %%); this run's vendor GEMM %.2f vs %.2f PFLOP/s here at 4096 rows, %.2f vs %.2f at 2048 | 69.8-70.4k / 82.1k / 26.2k |" % (" / ".join("%.1fk" % (t(k) / 1e3) for k in ("prefill_f16_b1_s2048", "prefill_f16_b8_s512", "prefill_f16_b1_s128")), f("prefill_f16_b1_s2048"), f("prefill_f16_b8_s512"), f("prefill_f16_b1_s128"), 100 * d["roofline_prefill"]["frac_event"], p["measured_vendor_gemm_f16_TFLOPs"] / 1e3, p["llmie_gemm_f16_TFLOPs"] / 1e3, p["measured_vendor_gemm_f16_TFLOPs_2048tok"] / 1e3, p["llmie_gemm_f16_TFLOPs_2048tok"] / 1e3))
rows.append("| fp8 prefill b8 s512 / b1 s2048 (configs[4]) | **%s tok/s** | %.0f / %.0f %% of 5 PFLOP/s; e4m3 GEMM alone %.2f PFLOP/s at 4096 x 12288 x 4096 (2.17-2.26 over the runs) | 130.6k / 108.7k |" % (" / ".join("%.1fk" % (t(k) / 1e3) for k in ("prefill_fp8_b8_s512", "prefill_fp8_b1_s2048")), f("prefill_fp8_b8_s512"), f("prefill_fp8_b1_s2048"), p["llmie_gemm_fp8_TFLOPs"] / 1e3))
cb = d["cpu_baseline"]
rows.append("| CPU baseline (oracle, OpenMP team = the box's CPU share, %d threads) | %.1f tok/s (one thread: %.2f) | -- | 5.2 |" % (cb["cores"], cb["value"], cb["one_thread"]["value"]))
table = "| config | result (round 2, `profiles/r02_bench_final.json`) | fraction of roofline | round 1 |\n|---|---|---|---|\n" + "\n".join(rows) + "\n"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("| config | result (round 2")
b = s.index("History of the headline on this hardware:")
open(path, "w").write(s[:a] + table + "\n" + s[b:])
print("rewrote the results table")
