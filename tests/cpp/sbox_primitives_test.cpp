// Host unit test of sda_amd/csrc/sbox_primitives.hpp (the same header the device kernels compile): reads commands
// with hex operands on stdin, prints hex results; tests/test_sealedbox_cpu.py compares them with the published
// vectors and with oracle/sealedbox_oracle.py.
//   x25519 <k:32> <u:32>          hsalsa <key:32> <in:16>       salsa <key:32> <nonce:8> <counter decimal>
//   nonce <epk:32> <pk:32>        poly <key:32> <msg:any>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <iostream>
#include <string>
#include <vector>

#include "../../sda_amd/csrc/sbox_primitives.hpp"

using namespace sda::sbx;

static std::vector<uint8_t> unhex(const std::string& s) {
    std::vector<uint8_t> out;
    if (s == "-") return out;
    for (size_t i = 0; i + 1 < s.size(); i += 2) out.push_back((uint8_t)strtoul(s.substr(i, 2).c_str(), nullptr, 16));
    return out;
}
static void words(const std::vector<uint8_t>& b, uint32_t* w, size_t n) {
    for (size_t i = 0; i < n; ++i) w[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
}
static void put(const uint32_t* w, size_t n) {
    for (size_t i = 0; i < n; ++i) printf("%02x%02x%02x%02x", w[i] & 255, (w[i] >> 8) & 255, (w[i] >> 16) & 255, w[i] >> 24);
    printf("\n");
}

int main() {
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd == "x25519") {
            std::string a, b; std::cin >> a >> b;
            uint32_t k[8], u[8], o[8];
            words(unhex(a), k, 8); words(unhex(b), u, 8);
            x25519(o, k, u); put(o, 8);
        } else if (cmd == "hsalsa") {
            std::string a, b; std::cin >> a >> b;
            uint32_t k[8], in[4], o[8];
            words(unhex(a), k, 8); words(unhex(b), in, 4);
            hsalsa20(o, k, in); put(o, 8);
        } else if (cmd == "salsa") {
            std::string a, b; unsigned long long ctr; std::cin >> a >> b >> ctr;
            uint32_t k[8], n[2], o[16];
            words(unhex(a), k, 8); words(unhex(b), n, 2);
            salsa20_block(o, k, n[0], n[1], ctr); put(o, 16);
        } else if (cmd == "nonce") {
            std::string a, b; std::cin >> a >> b;
            uint32_t e[8], p[8], o[6];
            words(unhex(a), e, 8); words(unhex(b), p, 8);
            seal_nonce(o, e, p); put(o, 6);
        } else if (cmd == "poly") {
            std::string a, b; std::cin >> a >> b;
            std::vector<uint8_t> key = unhex(a), msg = unhex(b);
            uint32_t kw[8];
            words(key, kw, 8);
            P26 r, h, c, t;
            p26_clamped_r(r, kw);
            for (int i = 0; i < 5; ++i) h.v[i] = 0;
            for (size_t off = 0; off < msg.size(); off += 16) {
                const size_t n = msg.size() - off < 16 ? msg.size() - off : 16;
                uint8_t buf[16] = {0};
                memcpy(buf, msg.data() + off, n);
                uint32_t w[4];
                words(std::vector<uint8_t>(buf, buf + 16), w, 4);
                p26_from_piece(c, w, (uint32_t)n);
                p26_add(t, h, c);
                p26_mul(h, t, r);
            }
            uint32_t tag[4];
            p26_finish(tag, h, kw + 4); put(tag, 4);
        } else {
            fprintf(stderr, "unknown command %s\n", cmd.c_str());
            return 2;
        }
    }
    return 0;
}
