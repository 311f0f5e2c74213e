// CPU-only robustness harness for the two parsers that read untrusted files (PLY, SPZ): built with
// -fsanitize=address,undefined from the product's own gs_ply.cpp / gs_spz.cpp and fed truncated and
// bit-flipped variants of valid inputs.  Any status is fine; a crash, an out-of-bounds access or UB
// is not.  (GPU AddressSanitizer is not available on the pool: sanitizers run on the CPU build.)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <vector>

#include "../../include/gs3d.h"

// the product's gs_fail lives in the HIP translation unit; the parsers only need it to return the code
extern "C++" gs_status gs_fail(gs_status code, uint64_t, uint64_t, uint64_t, const char *, ...) { return code; }

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void try_ply(const std::vector<uint8_t> &b) {
    size_t n = 0;
    int32_t inria = 0;
    if (gs_ply_read(b.data(), b.size(), nullptr, 0, &n, &inria) != GS_OK) return;
    if (n > (1u << 20)) return;   // a header may legitimately announce more than the body holds: bounded here
    std::vector<gs_ply_gaussian_pod> pods(n ? n : 1);
    (void)gs_ply_read(b.data(), b.size(), pods.data(), n, &n, &inria);
}

static void try_spz(const std::vector<uint8_t> &b, bool gz) {
    size_t n = 0;
    gs_spz_header h;
    auto fn = gz ? gs_spz_decode : gs_spz_decode_decompressed;
    if (fn(b.data(), b.size(), &h, nullptr, 0, &n) != GS_OK) return;
    if (n > (1u << 20)) return;
    std::vector<gs_gaussian> g(n ? n : 1);
    (void)fn(b.data(), b.size(), &h, g.data(), n, &n);
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    std::vector<uint8_t> ply = slurp(argv[1]), spz = slurp(argv[2]);
    const int rounds = std::atoi(argv[3]);
    if (ply.empty() || spz.empty()) return 2;
    // the decompressed SPZ payload as a third corpus
    std::vector<uint8_t> raw;
    {
        size_t n = 0;
        if (gs_spz_decompress(spz.data(), spz.size(), nullptr, 0, &n) != GS_OK) return 3;
        raw.resize(n);
        if (gs_spz_decompress(spz.data(), spz.size(), raw.data(), raw.size(), &n) != GS_OK) return 3;
    }
    // an ascii PLY too (custom property order path): first 3 vertices of the file re-emitted as text
    std::vector<uint8_t> ascii;
    {
        size_t n = 0;
        int32_t inria = 0;
        gs_ply_read(ply.data(), ply.size(), nullptr, 0, &n, &inria);
        std::vector<gs_ply_gaussian_pod> pods(n);
        gs_ply_read(ply.data(), ply.size(), pods.data(), n, &n, &inria);
        std::string t = "ply\nformat ascii 1.0\nelement vertex 3\n";
        for (uint32_t i = 0; i < 62; i++) t += std::string("property float ") + gs_ply_property_name(i) + "\n";
        t += "end_header\n";
        for (int v = 0; v < 3 && v < (int)n; v++) {
            const float *f = (const float *)&pods[v];
            for (int k = 0; k < 62; k++) { char buf[32]; std::snprintf(buf, sizeof buf, "%g ", f[k]); t += buf; }
            t += "\n";
        }
        ascii.assign(t.begin(), t.end());
    }
    std::mt19937 rng(12345);
    std::vector<uint8_t> *corp[4] = {&ply, &spz, &raw, &ascii};
    long runs = 0;
    for (int c = 0; c < 4; c++) {
        const std::vector<uint8_t> &base = *corp[c];
        // every truncation of the first 600 bytes, then a stride through the rest
        for (size_t len = 0; len <= base.size(); len += (len < 600 ? 1 : 97)) {
            std::vector<uint8_t> b(base.begin(), base.begin() + len);
            if (c == 0 || c == 3) try_ply(b); else try_spz(b, c == 1);
            runs++;
        }
        for (int r = 0; r < rounds; r++) {
            std::vector<uint8_t> b = base;
            int flips = 1 + (int)(rng() % 8);
            for (int k = 0; k < flips; k++) {
                size_t pos = rng() % b.size();
                switch (rng() % 4) {
                case 0: b[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: b[pos] = (uint8_t)rng(); break;
                case 2: b[pos] = 0xff; break;
                default: b[pos] = (uint8_t)('0' + rng() % 10); break;
                }
            }
            if (rng() % 4 == 0) b.resize(rng() % (b.size() + 1));
            if (c == 0 || c == 3) try_ply(b); else try_spz(b, c == 1);
            runs++;
        }
    }
    // encoders with hostile values
    gs_gaussian g[3];
    std::memset(g, 0, sizeof g);
    const float bad[] = {0.0f, -0.0f, 1e38f, -1e38f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1e-45f};
    for (float v : bad) {
        for (auto &x : g) {
            for (float &p : x.pos) p = v;
            for (float &p : x.rot) p = v;
            for (float &p : x.scale) p = v;
            for (float &p : x.sh) p = v;
        }
        for (uint32_t ver = 1; ver <= 3; ver++) {
            gs_spz_options o;
            gs_spz_options_default(&o);
            o.version = ver;
            size_t n = 0;
            if (gs_spz_encode(g, 3, &o, nullptr, 0, &n) == GS_OK) {
                std::vector<uint8_t> out(n + 16);
                (void)gs_spz_encode(g, 3, &o, out.data(), out.size(), &n);
                out.resize(n);
                try_spz(out, true);
            }
            runs++;
        }
        gs_ply_gaussian_pod pp[3];
        gs_gaussian_to_ply(g, 3, pp);
        gs_gaussian back[3];
        gs_gaussian_from_ply(pp, 3, back);
    }
    std::printf("fuzz OK: %ld inputs\n", runs);
    return 0;
}
