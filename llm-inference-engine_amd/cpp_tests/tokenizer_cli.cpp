// Host-only driver for the tokenizer parity test (tests/test_tokenizer_cpu.py): g++ -std=c++17, no GPU.
//   tokenizer_cli <vocab file>   reads lines "E <hex bytes of the text>" or "D id id ..." from stdin, prints the ids / hex of the text
#include <iostream>
#include <sstream>

#include "../src/models/tokenizer.h"

static std::string unhex(const std::string &h) {
    std::string s;
    for (size_t i = 0; i + 1 < h.size(); i += 2) s.push_back(static_cast<char>(std::stoi(h.substr(i, 2), nullptr, 16)));
    return s;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    Tokenizer tok;
    tok.Initialize(argv[1]);
    if (!tok.loaded) return 3;
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream in(line.substr(1));
        if (line[0] == 'E') {
            std::string hex;
            in >> hex;
            for (int id : tok.Encode(unhex(hex))) std::cout << id << ' ';
            std::cout << '\n';
        } else {
            std::vector<int> ids;
            for (int id; in >> id;) ids.push_back(id);
            static const char *hx = "0123456789abcdef";
            for (unsigned char c : tok.Decode(ids)) std::cout << hx[c >> 4] << hx[c & 15];
            std::cout << '\n';
        }
    }
    return 0;
}
