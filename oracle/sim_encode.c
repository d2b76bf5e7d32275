// Development aid: CPU simulation of the wave-parallel match finder in
// streamly-lz4_amd/csrc/encode_wave.hpp (sizes only), to tune ratio without a GPU.
// Build: gcc -O2 -I oracle oracle/sim_encode.c oracle/lz4_oracle.c -o /tmp/sim_encode
#include "lz4_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int HLOG = 12;
static uint32_t hash5(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - HLOG)); }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

static int VARIANT = 0;

static int sim(const uint8_t *src, int n, int accel)
{
    static uint32_t table[4096];
    memset(table, 0, sizeof(table));
    long out = 0; int anchor = 0;
    if (n == 0) return 1;
    if (n >= 13) {
        int mfl = n - 11, matchlimit = n - 5;
        uint32_t miss0 = (uint32_t)accel << 6, missAcc = miss0;
        long p = 0;
        while (p < mfl) {
            long step = missAcc >> 6;
            int first = 64; long pos[64]; uint32_t cand[64], h[64]; int valid[64];
            for (int l = 0; l < 64; l++) {
                long off = (VARIANT & 1) ? (l < 3 ? l : 2 + (l - 2) * step) : l * step;   // variant 1: p, p+1, p+2, p+2+step...
                pos[l] = p + off; valid[l] = pos[l] < mfl;
                if (!valid[l]) continue;
                h[l] = hash5(src + pos[l]); cand[l] = table[h[l]];
                if (first == 64 && cand[l] < pos[l] && pos[l] - cand[l] <= 65535 && rd32(src + cand[l]) == rd32(src + pos[l])) first = l;
            }
            for (int l = 0; l < 64; l++) if (valid[l] && l <= first) table[h[l]] = (uint32_t)pos[l];
            if (first == 64) { missAcc += 64; p = pos[63] + step; if (!(VARIANT&1)) p = p; continue; }
            int mpos = (int)pos[first], cpos = (int)cand[first];
            while (mpos > anchor && cpos > 0 && src[mpos - 1] == src[cpos - 1]) { mpos--; cpos--; }
            int ml = 4; while (mpos + ml < matchlimit && src[mpos + ml] == src[cpos + ml]) ml++;
            int lit = mpos - anchor, mc = ml - 4;
            out += 1 + lit + 2 + (lit >= 15 ? (lit - 15) / 255 + 1 : 0) + (mc >= 15 ? (mc - 15) / 255 + 1 : 0);
            anchor = mpos + ml; p = anchor; missAcc = miss0;
            if (VARIANT & 2) { if (anchor - 2 >= 0 && anchor < mfl) table[hash5(src + anchor - 2)] = anchor - 2; }
        }
    }
    int last = n - anchor;
    out += 1 + last + (last >= 15 ? (last - 15) / 255 + 1 : 0);
    return (int)out;
}

// multi-match per window (encode_wave v2): every hit lane extends its own match; matches are then
// selected greedily left to right; uncovered probed positions before the last selected end are inserted.
static int RUNHEAD = 0, NOEND2 = 0, BACKMAX = 1000;
static int sim_multi(const uint8_t *src, int n, int accel)
{
    static uint32_t table[4096];
    memset(table, 0, sizeof(table));
    long out = 0; int anchor = 0;
    if (n == 0) return 1;
    if (n >= 13) {
        int mfl = n - 11, matchlimit = n - 5;
        uint32_t miss0 = (uint32_t)accel << 6, missAcc = miss0;
        long p = 0;
        while (p < mfl) {
            long step = missAcc >> 6;
            long pos[64]; uint32_t cand[64], h[64]; int valid[64], hit[64], ml[64];
            for (int l = 0; l < 64; l++) {
                long off = (l < 3 ? l : 2 + (l - 2) * step);
                pos[l] = p + off; valid[l] = pos[l] < mfl; hit[l] = 0;
                if (!valid[l]) continue;
                h[l] = hash5(src + pos[l]); cand[l] = table[h[l]];
                if (cand[l] < pos[l] && pos[l] - cand[l] <= 65535 && rd32(src + cand[l]) == rd32(src + pos[l])) {
                    hit[l] = 1; int m = 4; while (pos[l] + m < matchlimit && src[pos[l] + m] == src[cand[l] + m]) m++; ml[l] = m;
                }
            }
            if (RUNHEAD && step == 1) {
                // only run heads (candidate not consecutive with the left neighbour's) verify; the rest derive
                int head = 0;
                for (int l = 0; l < 64; l++) {
                    if (!valid[l]) { hit[l] = 0; continue; }
                    int contin = l > 0 && valid[l-1] && cand[l] == cand[l-1] + 1;
                    if (!contin) { head = l; continue; }
                    // derive from head
                    if (hit[head] && ml[head] - (l - head) >= 4) { hit[l] = 1; ml[l] = ml[head] - (l - head); }
                    else hit[l] = 0;
                }
            }
            long pEnd = anchor; int nsel = 0; long lastEnd = -1; int covered[64]; memset(covered, 0, sizeof(covered));
            for (int l = 0; l < 64; l++) {
                if (!hit[l] || pos[l] < pEnd) continue;
                int mpos = (int)pos[l], cpos = (int)cand[l], m = ml[l];
                { int bk = 0; while (bk < BACKMAX && mpos > pEnd && cpos > 0 && src[mpos - 1] == src[cpos - 1]) { mpos--; cpos--; m++; bk++; } }
                int lit = mpos - (int)pEnd, mc = m - 4;
                out += 1 + lit + 2 + (lit >= 15 ? (lit - 15) / 255 + 1 : 0) + (mc >= 15 ? (mc - 15) / 255 + 1 : 0);
                for (int q = 0; q < 64; q++) if (valid[q] && pos[q] > pos[l] && pos[q] < pos[l] + ml[l]) covered[q] = 1;
                pEnd = pos[l] + ml[l]; lastEnd = pEnd; nsel++;
            }
            for (int l = 0; l < 64; l++) if (valid[l] && !covered[l] && (nsel == 0 || pos[l] < lastEnd)) table[h[l]] = (uint32_t)pos[l];
            if (nsel == 0) { missAcc += 64; p += 2 + 62 * step; continue; }
            // ip-2 insertion for every selected match end (in order)
            if (!NOEND2) {
                long pe = anchor;
                for (int l = 0; l < 64; l++) {
                    if (!hit[l] || pos[l] < pe) continue;
                    pe = pos[l] + ml[l];
                    if (pe < mfl) table[hash5(src + pe - 2)] = (uint32_t)(pe - 2);
                }
            }
            anchor = (int)lastEnd; p = lastEnd; missAcc = miss0;
        }
    }
    int last = n - anchor;
    out += 1 + last + (last >= 15 ? (last - 15) / 255 + 1 : 0);
    return (int)out;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "lzsynth";
    int bl = argc > 2 ? atoi(argv[2]) : 65536, nb = argc > 3 ? atoi(argv[3]) : 16;
    uint8_t *buf = malloc(bl), *dst = malloc(bl + bl / 255 + 64);
    int accels[] = {1, 2, 5, 20, 400};
    for (int ai = 0; ai < 5; ai++) {
        long ref = 0, s[4] = {0, 0, 0, 0}, sm = 0, sm2 = 0, sm3 = 0, sm4 = 0, sm5 = 0, sm6 = 0;
        for (int b = 0; b < nb; b++) {
            if (!strcmp(kind, "lzsynth")) orc_gen_lzsynth(buf, bl, b, 16, 2048);
            else if (!strcmp(kind, "text")) orc_gen_text(buf, bl, b);
            else orc_gen_random(buf, bl, b);
            ref += orc_compress_block(buf, dst, bl, bl + bl / 255 + 16, accels[ai]);
            for (VARIANT = 0; VARIANT < 4; VARIANT++) s[VARIANT] += sim(buf, bl, accels[ai]);
            RUNHEAD = 0; sm += sim_multi(buf, bl, accels[ai]);
            RUNHEAD = 1; sm2 += sim_multi(buf, bl, accels[ai]);
            NOEND2 = 1; sm3 += sim_multi(buf, bl, accels[ai]); BACKMAX = 8; sm4 += sim_multi(buf, bl, accels[ai]);
            HLOG = 11; sm5 += sim_multi(buf, bl, accels[ai]); HLOG = 10; sm6 += sim_multi(buf, bl, accels[ai]); HLOG = 12; BACKMAX = 1000; NOEND2 = 0;
        }
        printf("%s accel %3d: ref %.4f | v0 %.4f v1 %.4f v2 %.4f v3 %.4f multi %.4f runhead %.4f noend2 %.4f +back8 %.4f hlog11 %.4f hlog10 %.4f (ratio)\n", kind, accels[ai],
               (double)nb * bl / ref, (double)nb * bl / s[0], (double)nb * bl / s[1], (double)nb * bl / s[2], (double)nb * bl / s[3], (double)nb * bl / sm, (double)nb * bl / sm2, (double)nb * bl / sm3, (double)nb * bl / sm4, (double)nb * bl / sm5, (double)nb * bl / sm6);
    }
    return 0;
}
