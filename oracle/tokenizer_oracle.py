"""CPU restatement of the reference tokenizer (TEST INFRASTRUCTURE ONLY, like the rest of oracle/): byte trie +
priority-queue merges, following src/models/tokenizer.h line by line -- Initialize :138-167, TryMergePairs :168-186,
Encode :188-293, DecodeTokens :305-347.  Parity unpinned: the reference holds no tokenizer test or fixture and no
vocabulary file, so this restatement (reviewed against the source) is the only checker of llm-inference-engine_amd/api/tokenizer.hpp,
on synthetic vocabularies."""
import heapq
import struct

NONE = -999999
BLANK = b"\xe2\x96\x81"


class _Node:
    __slots__ = ("token", "score", "next")

    def __init__(self):
        self.token, self.score, self.next = NONE, 0.0, {}


class RefTokenizer:
    def __init__(self):
        self.root = _Node()
        self.text_of, self.id_of = {}, {}

    def insert(self, b, tid, score):  # tokenizer.h:124-137
        n = self.root
        for c in b:
            n = n.next.setdefault(c, _Node())
        n.token, n.score = tid, score
        self.text_of[tid], self.id_of[b] = b, tid

    def load(self, path):  # tokenizer.h:138-167
        data = open(path, "rb").read()
        off = [0]

        def i32():
            v = struct.unpack_from("<i", data, off[0])[0]
            off[0] += 4
            return v

        def string():
            n = i32()
            off[0] += n
            return data[off[0] - n:off[0]]

        if i32() >= 1:
            for _ in range(i32()):
                string()
                string()
        for _ in range(i32()):
            b = bytes(i32() & 0xFF for _ in range(i32()))
            tid = i32()
            score = struct.unpack_from("<f", data, off[0])[0]
            off[0] += 4
            self.insert(b, tid, score)

    def encode(self, text):  # tokenizer.h:188-293
        fix = b"<FLM_FIX_TOKEN_"
        s = b"" if (len(text) > 15 and text[:15] == fix) else BLANK
        for i, c in enumerate(text):
            if c == 0x20:
                if i != 0 and text[i - 1] != 0x20:
                    s += BLANK
            else:
                s += bytes([c])
        sym = []  # [node, pos, len, prev, next, fix]
        i = 0
        while i < len(s):
            if i + 15 < len(s) and s[i:i + 15] == fix:
                i += 15
                now = 0
                while i < len(s) and 0x30 <= s[i] <= 0x39:
                    now = now * 10 + s[i] - 0x30
                    i += 1
                sym.append([None, i, 0, len(sym) - 1, len(sym) + 1, now])
                i += 1
                continue
            node, pos = self.root, i - 1
            for j in range(i, len(s)):
                if s[j] in node.next:
                    node = node.next[s[j]]
                    if node.token != NONE:
                        pos = j
                        break
                else:
                    break
            if pos >= i:
                sym.append([node, i, pos - i + 1, len(sym) - 1, len(sym) + 1, NONE])
                i = pos
            else:
                sym.append([None, i, 0, len(sym) - 1, len(sym) + 1, NONE])
            i += 1
        if not sym:
            return []
        sym[-1][4] = -1
        heap = []

        def try_merge(l, r):  # tokenizer.h:168-186
            if l == -1 or r == -1 or sym[l][2] == 0 or sym[r][2] == 0:
                return
            node = sym[l][0]
            for k in range(sym[r][1], sym[r][1] + sym[r][2]):
                if s[k] in node.next:
                    node = node.next[s[k]]
                else:
                    return
            if node.token == NONE:
                return
            # std::priority_queue pops the largest: higher score, then smaller l  (operator< :95-97)
            heapq.heappush(heap, (-node.score, l, r, sym[l][2] + sym[r][2]))

        for k in range(1, len(sym)):
            try_merge(k - 1, k)
        while heap:
            _, l, r, size = heapq.heappop(heap)
            if sym[l][2] == 0 or sym[r][2] == 0 or sym[l][2] + sym[r][2] != size:
                continue
            for k in range(sym[r][1], sym[r][1] + sym[r][2]):
                sym[l][0] = sym[l][0].next[s[k]]
            sym[l][2] += sym[r][2]
            sym[r][2] = 0
            sym[l][4] = sym[r][4]
            if sym[r][4] >= 0:
                sym[sym[r][4]][3] = l
            try_merge(sym[l][3], l)
            try_merge(l, sym[l][4])
        out = []
        for node, pos, ln, _, _, fixid in sym:
            if ln > 0:
                out.append(node.token)
            elif node is None:
                if fixid != NONE:
                    out.append(fixid)
                else:
                    name = b"<0x%02X>" % s[pos]
                    if name in self.id_of:
                        out.append(self.id_of[name])
        return out

    def decode(self, ids):  # tokenizer.h:305-347
        ret = b""
        for t in ids:
            s = self.text_of.get(t, b"")
            if len(s) == 6 and s[:3] == b"<0x" and s[-1:] == b">":
                s = bytes([int(s[3:5], 16)])
            if s == b"<n>":
                ret += b"\n"
            elif s == b"<|tab|>":
                ret += b"\t"
            else:
                ret += s
        ret = ret.replace(BLANK, b" ")
        if ret.find(b"<|blank_") != -1 and len(ret) >= 10:
            return b" " * int(ret[8:len(ret) - 2] or b"0")
        return ret
