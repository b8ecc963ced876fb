// C++ API mirror, part 7b: Tokenizer (src/models/tokenizer.h:56-347).  Host-only, no HIP dependency.
//
// Same vocabulary file, same Encode / Decode results as the reference's trie + priority-queue tokenizer, built
// differently: the vocabulary is one hash map from token bytes to (id, score) -- a merge candidate is simply the
// concatenation of two adjacent spans of the normalised text looked up in that map (the reference walks a byte trie to the
// same node), and the initial symbols are the SHORTEST vocabulary prefix at each position (the reference's trie walk
// stops at the first node that carries a token, tokenizer.h:222-246).
//   file (tokenizer.h:138-167): int32 version; if version >= 1: int32 n, n x (string key, string value) with
//   string = int32 len + bytes; int32 vocab; vocab x { int32 len; len x int32 (one byte value each); int32 id; float score }
//   Encode (tokenizer.h:188-293): text -> U+2581 + text with every run of spaces collapsed to one U+2581 (leading spaces
//   dropped), "<FLM_FIX_TOKEN_n>" passes id n through, greedy merges by descending score (ties: leftmost pair first),
//   bytes without a token fall back to "<0xNN>" when the vocabulary has it (dropped otherwise).
//   Decode (tokenizer.h:305-347): "<0xNN>" -> byte, "<n>" -> newline, "<|tab|>" -> tab, U+2581 -> space,
//   "<|blank_k|>" -> k spaces.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

class Tokenizer {
    struct Entry {
        int id;
        float score;
    };
    std::unordered_map<std::string, Entry> vocab_;
    std::unordered_map<int, std::string> text_of_;
    size_t max_len_ = 0;

    static std::string blank() { return std::string("\xE2\x96\x81"); }  // U+2581

    struct Span {  // a symbol = bytes [pos, pos + len) of the normalised text; len == 0: merged away or no token
        int pos, len, prev, next;
        bool known;  // false: a byte no token starts with (or a fixed id)
        int fixed;   // id given by <FLM_FIX_TOKEN_n>, else -1
    };
    struct Cand {
        float score;
        int l, r, size;
    };
    struct CandLess {  // max-heap: higher score first, then the leftmost pair (tokenizer.h:95-97)
        bool operator()(const Cand &a, const Cand &b) const { return a.score < b.score || (a.score == b.score && a.l > b.l); }
    };

public:
    bool loaded = false;
    std::string path;

    void Insert(const std::string &bytes, int id, float score) {
        vocab_[bytes] = Entry{id, score};
        text_of_[id] = bytes;
        if (bytes.size() > max_len_) max_len_ = bytes.size();
    }
    size_t size() const { return vocab_.size(); }

    void Initialize(const std::string &file) {
        path = file;
        loaded = false;
        FILE *f = std::fopen(file.c_str(), "rb");
        if (!f) {
            std::cerr << "[llmie] tokenizer file " << file << " not found: token ids are printed as <id>\n";
            return;
        }
        bool ok = true;
        auto rd_i32 = [&]() {
            int32_t v = 0;
            if (std::fread(&v, 4, 1, f) != 1) ok = false;
            return v;
        };
        auto rd_str = [&]() {
            const int32_t n = rd_i32();
            std::string s(ok && n > 0 ? static_cast<size_t>(n) : 0, '\0');
            if (ok && n > 0 && std::fread(&s[0], 1, static_cast<size_t>(n), f) != static_cast<size_t>(n)) ok = false;
            return s;
        };
        const int32_t version = rd_i32();
        if (ok && version >= 1) {
            const int32_t n = rd_i32();
            for (int32_t i = 0; ok && i < n; ++i) {
                (void)rd_str();
                (void)rd_str();
            }
        }
        const int32_t count = rd_i32();
        for (int32_t i = 0; ok && i < count; ++i) {
            const int32_t len = rd_i32();
            std::string bytes;
            for (int32_t j = 0; ok && j < len; ++j) bytes.push_back(static_cast<char>(rd_i32()));
            const int32_t id = rd_i32();
            float score = 0.f;
            if (std::fread(&score, 4, 1, f) != 1) ok = false;
            if (ok) Insert(bytes, id, score);
        }
        std::fclose(f);
        if (!ok) std::cerr << "[llmie] tokenizer file " << file << " is truncated\n";
        loaded = ok && !vocab_.empty();
    }

    std::vector<int> Encode(const std::string &text) const {
        if (!loaded)  // no vocabulary: the prompt ids the reference hard-codes (llama.cpp:328,340)
            return {1, 18637, 29892, 526, 366, 19861, 29973, 1815, 366, 5193, 304, 592, 29973};
        static const std::string fix = "<FLM_FIX_TOKEN_";
        std::string s = (text.size() > fix.size() && text.compare(0, fix.size(), fix) == 0) ? std::string() : blank();
        for (size_t i = 0; i < text.size(); ++i) {
            if (text[i] == ' ') {
                if (i != 0 && text[i - 1] != ' ') s += blank();
            } else {
                s += text[i];
            }
        }
        const int n = static_cast<int>(s.size());
        std::vector<Span> sym;
        for (int i = 0; i < n; ++i) {
            const int idx = static_cast<int>(sym.size());
            if (i + static_cast<int>(fix.size()) < n && s.compare(static_cast<size_t>(i), fix.size(), fix) == 0) {
                i += static_cast<int>(fix.size());
                int id = 0;
                while (i < n && s[i] >= '0' && s[i] <= '9') id = id * 10 + (s[i++] - '0');
                sym.push_back(Span{i, 0, idx - 1, idx + 1, false, id});  // i now sits on the closing '>' and skips it
                continue;
            }
            int len = 0;
            for (size_t l = 1; l <= max_len_ && i + static_cast<int>(l) <= n; ++l)
                if (vocab_.count(s.substr(static_cast<size_t>(i), l))) {
                    len = static_cast<int>(l);
                    break;
                }
            sym.push_back(Span{i, len, idx - 1, idx + 1, len > 0, -1});
            if (len > 0) i += len - 1;
        }
        if (sym.empty()) return {};
        sym.back().next = -1;

        std::priority_queue<Cand, std::vector<Cand>, CandLess> heap;
        auto consider = [&](int l, int r) {
            if (l < 0 || r < 0 || sym[l].len == 0 || sym[r].len == 0) return;
            const auto it = vocab_.find(s.substr(static_cast<size_t>(sym[l].pos), static_cast<size_t>(sym[l].len + sym[r].len)));
            if (it != vocab_.end()) heap.push(Cand{it->second.score, l, r, sym[l].len + sym[r].len});
        };
        for (int i = 1; i < static_cast<int>(sym.size()); ++i) consider(i - 1, i);
        while (!heap.empty()) {
            const Cand c = heap.top();
            heap.pop();
            if (sym[c.l].len == 0 || sym[c.r].len == 0 || sym[c.l].len + sym[c.r].len != c.size) continue;  // stale
            sym[c.l].len += sym[c.r].len;
            sym[c.r].len = 0;
            sym[c.l].next = sym[c.r].next;
            if (sym[c.r].next >= 0) sym[sym[c.r].next].prev = c.l;
            consider(sym[c.l].prev, c.l);
            consider(c.l, sym[c.l].next);
        }
        std::vector<int> ids;
        for (const Span &p : sym) {
            if (p.len > 0) {
                ids.push_back(vocab_.at(s.substr(static_cast<size_t>(p.pos), static_cast<size_t>(p.len))).id);
            } else if (!p.known) {
                if (p.fixed >= 0) {
                    ids.push_back(p.fixed);
                } else {
                    static const char *hex = "0123456789ABCDEF";
                    const unsigned c = static_cast<unsigned char>(s[static_cast<size_t>(p.pos)]);
                    const std::string fallback = std::string("<0x") + hex[c >> 4] + hex[c & 15] + ">";
                    const auto it = vocab_.find(fallback);
                    if (it != vocab_.end()) ids.push_back(it->second.id);
                }
            }
        }
        return ids;
    }

    std::string Decode(const std::vector<int> &ids) const {
        if (!loaded) {
            std::string s;
            for (int id : ids) s += "<" + std::to_string(id) + ">";
            return s;
        }
        std::string out;
        for (int id : ids) {
            const auto it = text_of_.find(id);
            std::string t = it == text_of_.end() ? std::string() : it->second;
            if (t.size() == 6 && t.compare(0, 3, "<0x") == 0 && t.back() == '>') {
                auto nib = [](char ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch - 'A' + 10; };
                t = std::string(1, static_cast<char>(nib(t[3]) * 16 + nib(t[4])));
            }
            if (t == "<n>") out += "\n";
            else if (t == "<|tab|>") out += "\t";
            else out += t;
        }
        const std::string b = blank();
        for (size_t pos = out.find(b); pos != std::string::npos; pos = out.find(b)) out.replace(pos, b.size(), " ");
        if (out.find("<|blank_") != std::string::npos && out.size() >= 10)
            return std::string(static_cast<size_t>(std::atoi(out.substr(8, out.size() - 10).c_str())), ' ');
        return out;
    }
    std::string DecodeTokens(const std::vector<int> &ids) const { return Decode(ids); }
};
