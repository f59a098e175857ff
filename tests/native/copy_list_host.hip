// Walks gfcopy lists on the CPU (tests/test_copy_list_host.py): the descriptor builder's alignment choice and block shares, and the unit mapping of
// copy_list_kernel -- every byte of every descriptor's rows moved exactly once, nothing beside them touched -- for the shapes the library uses (full tables,
// strided rows of odd lengths, 1-byte status tables, many tiny tables behind one large one) and for seeded random lists.  No HIP call is made.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../ground-fusion_amd/csrc/gf_copy_list.hpp"

struct Case { size_t rows, pitch, used, src_shift, dst_shift; };

static int run(const std::vector<Case>& cases, unsigned seed, const char* what) {
    std::mt19937 rng(seed);
    std::vector<std::vector<unsigned char>> src(cases.size()), dst(cases.size()), want(cases.size());
    gfcopy::Builder B;
    for (size_t i = 0; i < cases.size(); i++) {
        const Case& c = cases[i];
        const size_t bytes = c.rows * c.pitch + 64;
        src[i].resize(bytes + 64); dst[i].resize(bytes + 64); want[i].resize(bytes + 64);
        for (auto& b : src[i]) b = (unsigned char)rng();
        for (auto& b : dst[i]) b = (unsigned char)rng();
        want[i] = dst[i];
        // the 64-byte aligned start of a vector's storage is not guaranteed: align by hand, then apply the case's shift
        unsigned char* s0 = src[i].data() + ((64 - (uintptr_t)src[i].data() % 64) % 64) + c.src_shift;
        unsigned char* d0 = dst[i].data() + ((64 - (uintptr_t)dst[i].data() % 64) % 64) + c.dst_shift;
        unsigned char* w0 = want[i].data() + (d0 - dst[i].data());
        for (size_t r = 0; r < c.rows; r++) memcpy(w0 + r * c.pitch, s0 + r * c.pitch, std::min(c.used, c.pitch));
        B.add(s0, d0, c.rows, c.pitch, c.used);
    }
    if (!B.ok) { printf("%s: builder refused the list\n", what); return 1; }
    B.run_on_host();
    unsigned blocks = 0;
    for (int k = 0; k < B.L.n; k++) {
        const gfcopy::Desc& D = B.L.e[k];
        if (D.blk0 != blocks || D.nblk < 1) { printf("%s: descriptor %d starts at block %u, expected %u\n", what, k, D.blk0, blocks); return 1; }
        blocks += D.nblk;
        const size_t al = (size_t)D.pitch | D.used | (size_t)(uintptr_t)D.src | (size_t)(uintptr_t)D.dst;
        if (al % D.esize) { printf("%s: descriptor %d: unit of %u bytes on a misaligned descriptor\n", what, k, D.esize); return 1; }
        const unsigned wider = D.esize == 1 ? 4 : 2 * D.esize;   // the kernel's units: 16, 8, 4 or 1 bytes
        if (D.esize < 16 && al % wider == 0) { printf("%s: descriptor %d: unit of %u bytes where %u would do\n", what, k, D.esize, wider); return 1; }
    }
    if (blocks != B.nblocks || blocks > gfcopy::kBlocks + (unsigned)B.L.n) { printf("%s: %u blocks against nblocks %u\n", what, blocks, B.nblocks); return 1; }
    for (size_t i = 0; i < cases.size(); i++)
        if (dst[i] != want[i]) {
            size_t at = 0; while (dst[i][at] == want[i][at]) at++;
            printf("%s: table %zu differs at byte %zu (rows %zu pitch %zu used %zu)\n", what, i, at, cases[i].rows, cases[i].pitch, cases[i].used);
            return 1;
        }
    printf("%s: %zu tables, %lld bytes, %u blocks: ok\n", what, cases.size(), B.bytes, B.nblocks);
    return 0;
}

int main() {
    int bad = 0;
    // a back-end upload in miniature: full tables (rows collapse to one), strided tables with odd used lengths (4-byte units), a large observation table
    bad |= run({{1, 4096, 4096, 0, 0}, {256, 600, 600, 0, 0}, {256, 1500 * 4, 1117 * 4, 0, 0}, {256, 1500 * 40, 1116 * 40, 0, 0}, {256, 900, 6 * 8 * 17, 0, 0}, {64, 8 * 467, 2 * 467 * 8, 0, 0}}, 1, "upload");
    // a tracker hand-over: byte tables, u16 tables, float2 tables of B x cap entries
    bad |= run({{1, 256 * 152 * 8, 256 * 152 * 8, 0, 0}, {1, 256 * 152, 256 * 152, 0, 0}, {1, 256 * 152, 256 * 152, 0, 0}, {1, 256 * 152 * 2, 256 * 152 * 2, 0, 0}, {1, 256 * 4, 256 * 4, 0, 0}}, 2, "hand-over");
    // misaligned ends: 1-, 2-, 4- and 8-byte alignment of source, destination, pitch or length each force the smaller unit
    bad |= run({{3, 37, 21, 0, 0}, {5, 64, 48, 1, 1}, {5, 64, 48, 2, 2}, {7, 96, 40, 4, 4}, {7, 96, 40, 8, 8}, {2, 128, 100, 0, 8}, {9, 16, 16, 0, 0}, {1, 1, 1, 3, 5}}, 3, "misaligned");
    // one large table in front of many tiny ones: every descriptor keeps at least one block
    {
        std::vector<Case> c{{1, 8u << 20, 8u << 20, 0, 0}};
        for (int i = 0; i < 40; i++) c.push_back({1, 1024, 1024, 0, 0});
        bad |= run(c, 4, "one large + forty small");
    }
    std::mt19937 rng(99);
    for (int t = 0; t < 30 && !bad; t++) {
        std::vector<Case> c;
        const int n = 1 + rng() % 12;
        for (int i = 0; i < n; i++) {
            const size_t g = 1u << (rng() % 5);                    // granularity 1 .. 16
            const size_t pitch = g * (1 + rng() % 300), used = std::min(pitch, g * (1 + rng() % 300)), rows = 1 + rng() % 40;
            c.push_back({rows, pitch, used, (rng() % 4) * g % 64, (rng() % 4) * g % 64});
        }
        char name[32]; snprintf(name, sizeof name, "random %d", t);
        bad |= run(c, 100 + t, name);
    }
    return bad;
}
