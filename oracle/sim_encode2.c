// Development aid (round 3): CPU simulation of the wave-parallel match finders, sizes + statistics only,
// to check a policy's compression ratio before it is written as a kernel.
//   mode cur   : streamly-lz4_amd/csrc/encode_wave.hpp as shipped in round 2 (4-bit tags, run heads, window restart)
//   mode pipe  : round-3 pipelined finder: fixed 64-position windows, every probed position inserted at probe
//                time and positions that end up covered by a selected match rolled back DEPTH-1 windows later
// Build: gcc -O2 -I oracle oracle/sim_encode2.c oracle/lz4_oracle.c -o /tmp/sim2
#include "lz4_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t hash16(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 20)); }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

static int TAGBITS = 4;        // bits of tag compared
static int DEPTH = 2;          // pipe: windows in flight
static int BACKCAP = 8;        // head measures up to 8 bytes before itself
static int WIN = 64;
static int HEADCAP = 1000;      // heads per 64-position window that get a group (the kernel: 32)
static int END2MIN = 0;        // ... only behind matches of at least this length
static int END2 = 0;           // cur: also register (match end - 2), as the reference does (:1146)
static int MINC = 0;           // cur: candidates below this position are not taken
static long g_windows, g_heads, g_seqs, g_hits, g_ext2, g_hist[65], g_known;

typedef struct { uint16_t pos; uint8_t tag; uint8_t used; } Ent;

static int count_fwd(const uint8_t *src, int a, int b, int limit)
{
    int m = 0;
    while (a + m < limit && src[a + m] == src[b + m]) m++;
    return m;
}

static long seq_size(int lit, int ml)
{
    int mc = ml - 4;
    return 1 + lit + 2 + (lit >= 15 ? (lit - 15) / 255 + 1 : 0) + (mc >= 15 ? (mc - 15) / 255 + 1 : 0);
}

// ------------------------------------------------------------------ current kernel
static int sim_cur(const uint8_t *src, int n, int accel)
{
    static Ent table[4096];
    memset(table, 0, sizeof(table));
    long out = 0; int anchor = 0;
    if (n == 0) return 1;
    const uint32_t tmask = (1u << TAGBITS) - 1u;
    if (n >= 13) {
        const int mfl = n - 11, matchlimit = n - 5;
        uint32_t miss0 = (uint32_t)accel << 6, missAcc = miss0;
        long p = 0;
        while (p < mfl) {
            long step = missAcc >> 6;
            if (step == 1) {
                int p0 = (int)p;
                int valid[128], candOk[128], hit[128], head[128], myHead[128]; uint32_t h[128], tg[128], cand[128], ml[128], hback[128];
                for (int l = 0; l < WIN; l++) {
                    int pos = p0 + l; valid[l] = pos < mfl; candOk[l] = hit[l] = head[l] = 0; ml[l] = 0; hback[l] = 0; cand[l] = 0;
                    if (!valid[l]) continue;
                    uint32_t hx = hash16(src + pos); h[l] = hx >> 8; tg[l] = (hx >> (8 - TAGBITS)) & tmask;
                    cand[l] = table[h[l]].pos;
                    candOk[l] = cand[l] < (uint32_t)pos && (table[h[l]].tag & tmask) == tg[l] && cand[l] >= (uint32_t)MINC;
                    if (!table[h[l]].used && cand[l] == 0 && pos > 0) candOk[l] = (0 == tg[l]);   // zeroed table: tag 0, pos 0
                }
                int hd = 0, nhd = 0;
                for (int l = 0; l < WIN; l++) {
                    int contin = candOk[l] && l > 0 && candOk[l - 1] && cand[l] == cand[l - 1] + 1;
                    head[l] = candOk[l] && !contin;
                    if (head[l]) {
                        hd = l; g_heads++; nhd++;
                        int pos = p0 + l;
                        if (nhd <= HEADCAP && rd32(src + pos) == rd32(src + cand[l])) {
                            hit[l] = 1;
                            int maxLen = matchlimit - pos;
                            int m = count_fwd(src, pos, cand[l], matchlimit);
                            ml[l] = m < maxLen ? m : maxLen;
                            int b = 0;
                            if (cand[l] >= 8) while (b < BACKCAP && b < (int)cand[l] && src[pos - 1 - b] == src[cand[l] - 1 - b]) b++;
                            hback[l] = b;
                        }
                    }
                    myHead[l] = hd;
                    if (contin) {
                        int m = (int)ml[hd] - (l - hd);
                        hit[l] = hit[hd] && m >= 4; ml[l] = hit[l] ? m : 0; hback[l] = hback[hd];
                    }
                }
                int any = 0; for (int l = 0; l < WIN; l++) any |= hit[l];
                g_windows++;
                if (!any) {
                    for (int l = 0; l < WIN; l++) if (valid[l]) { table[h[l]].pos = (uint16_t)(p0 + l); table[h[l]].tag = tg[l]; table[h[l]].used = 1; }
                    missAcc += WIN; p += WIN; continue;
                }
                int sel[128]; memset(sel, 0, sizeof(sel));
                int pEnd = anchor, lastEnd = anchor;
                {
                    int l = 0;
                    while (l < WIN) {
                        if (!hit[l]) { l++; continue; }
                        sel[l] = 1; int endk = p0 + l + (int)ml[l]; lastEnd = endk;
                        int sh = endk - p0; if (sh >= WIN) break;
                        l = sh;
                    }
                }
                int nextP = lastEnd > p0 + WIN ? lastEnd : p0 + WIN;
                int prevEnd = anchor;
                for (int l = 0; l < WIN; l++) {
                    int pos = p0 + l;
                    // covered: strictly inside a selected match
                    int covered = 0;
                    for (int q = l - 1; q >= 0; q--) if (sel[q]) { covered = pos < p0 + q + (int)ml[q]; break; }
                    int end2 = 0; if (END2) for (int q = l - 1; q >= 0; q--) if (sel[q]) { end2 = pos == p0 + q + (int)ml[q] - 2 && (int)ml[q] >= END2MIN; break; }
                    if (valid[l] && (!covered || end2) && pos < nextP) { table[h[l]].pos = (uint16_t)pos; table[h[l]].tag = tg[l]; table[h[l]].used = 1; }
                    if (sel[l]) {
                        int mstart = pos, mcand = (int)cand[l];
                        int room = mstart - prevEnd; if (mcand < room) room = mcand;
                        int back = (l - myHead[l]) + (int)hback[l]; if (room < back) back = room;
                        mstart -= back;
                        out += seq_size(mstart - prevEnd, pos + (int)ml[l] - mstart);
                        g_seqs++;
                        prevEnd = pos + (int)ml[l];
                    }
                }
                (void)pEnd;
                anchor = lastEnd; p = nextP; missAcc = miss0;
                continue;
            }
            // strided window: first match only
            int first = 64; long pos[64]; uint32_t cand[64], h[64], tg[64]; int valid[64];
            for (int l = 0; l < 64; l++) {
                pos[l] = p + (l < 3 ? l : 2 + (long)(l - 2) * step); valid[l] = pos[l] < mfl;
                if (!valid[l]) continue;
                uint32_t hx = hash16(src + pos[l]); h[l] = hx >> 8; tg[l] = (hx >> (8 - TAGBITS)) & tmask;
                cand[l] = table[h[l]].pos;
                int ok = cand[l] < pos[l] && (table[h[l]].tag & tmask) == tg[l];
                if (first == 64 && ok && rd32(src + cand[l]) == rd32(src + pos[l])) first = l;
            }
            for (int l = 0; l < 64; l++) if (valid[l] && l <= first) { table[h[l]].pos = (uint16_t)pos[l]; table[h[l]].tag = tg[l]; table[h[l]].used = 1; }
            if (first == 64) { if (missAcc < 0x7fffff00u) missAcc += 64; p += 2 + 62 * step; continue; }
            int mpos = (int)pos[first], cpos = (int)cand[first];
            while (mpos > anchor && cpos > 0 && src[mpos - 1] == src[cpos - 1]) { mpos--; cpos--; }
            int ml = 4 + count_fwd(src, mpos + 4, cpos + 4, matchlimit);
            out += seq_size(mpos - anchor, ml); g_seqs++;
            anchor = mpos + ml; p = anchor; missAcc = miss0;
            if (anchor < mfl) { uint32_t hx = hash16(src + anchor - 2); table[hx >> 8].pos = (uint16_t)(anchor - 2); table[hx >> 8].tag = (hx >> (8 - TAGBITS)) & tmask; table[hx >> 8].used = 1; }
        }
    }
    int last = n - anchor;
    out += 1 + last + (last >= 15 ? (last - 15) / 255 + 1 : 0);
    return (int)out;
}

// ------------------------------------------------------------------ pipelined finder
// Window w (positions [64w, 64w+64)) is PROBED at step w: every valid position reads its bucket (old entry kept) and
// writes itself.  It is FINISHED at step w + DEPTH - 1 (after the probes of windows up to w + DEPTH - 1): hits,
// lengths, greedy selection continuing from the running anchor, and every position of the window that lies strictly
// inside a selected match (of this or an earlier window) puts the old entry back if the bucket still holds it.
typedef struct { int valid[64], candOk[64]; uint32_t h[64], tg[64], cand[64]; Ent old[64]; int p0; } Win;
static int KNOWN = 1;          // positions below the anchor known at probe time are neither probed nor inserted
static int NOROLL = 0;         // 1: never roll back (insert-all policy)
static int sim_pipe(const uint8_t *src, int n, int accel)
{
    static Ent table[4096];
    memset(table, 0, sizeof(table));
    long out = 0; int anchor = 0;
    if (n == 0) return 1;
    const uint32_t tmask = (1u << TAGBITS) - 1u;
    (void)accel;
    if (n >= 13) {
        const int mfl = n - 11, matchlimit = n - 5;
        const int nw = (mfl + 63) / 64;
        Win *ring = calloc(DEPTH, sizeof(Win));
        for (int step = 0; step < nw + DEPTH - 1; step++) {
            if (step < nw) {
                Win *w = &ring[step % DEPTH]; w->p0 = step * 64;
                for (int l = 0; l < 64; l++) {
                    int pos = w->p0 + l; w->valid[l] = pos < mfl && (!KNOWN || pos >= anchor); w->candOk[l] = 0;
                    if (!w->valid[l]) continue;
                    uint32_t hx = hash16(src + pos); w->h[l] = hx >> 8; w->tg[l] = (hx >> (8 - TAGBITS)) & tmask;
                    w->old[l] = table[w->h[l]];
                    w->cand[l] = w->old[l].pos;
                    w->candOk[l] = w->old[l].used && w->cand[l] < (uint32_t)pos && (w->old[l].tag & tmask) == w->tg[l];
                }
                for (int l = 0; l < 64; l++) if (w->valid[l]) { Ent e = { (uint16_t)(w->p0 + l), (uint8_t)w->tg[l], 1 }; table[w->h[l]] = e; }
            }
            int fw = step - (DEPTH - 1);
            if (fw < 0) continue;
            Win *w = &ring[fw % DEPTH];
            g_windows++;
            int hit[64], myHead[64]; uint32_t ml[64], hback[64];
            int hd = 0, nh = 0;
            for (int l = 0; l < 64; l++) {
                hit[l] = 0; ml[l] = 0; hback[l] = 0;
                int contin = w->candOk[l] && l > 0 && w->candOk[l - 1] && w->cand[l] == w->cand[l - 1] + 1;
                if (w->candOk[l] && !contin) nh++;
                int head = w->candOk[l] && !contin;
                int pos = w->p0 + l;
                if (head) {
                    hd = l; g_heads++;
                    if (rd32(src + pos) == rd32(src + w->cand[l])) {
                        hit[l] = 1; g_hits++;
                        int maxLen = matchlimit - pos;
                        int m = count_fwd(src, pos, w->cand[l], matchlimit);
                        ml[l] = m < maxLen ? m : maxLen;
                        if ((int)ml[l] > 56) g_ext2++;
                        int b = 0;
                        while (b < BACKCAP && b < (int)w->cand[l] && b < pos && src[pos - 1 - b] == src[w->cand[l] - 1 - b]) b++;
                        hback[l] = b;
                    }
                }
                myHead[l] = hd;
                if (contin) {
                    int m = (int)ml[hd] - (l - hd);
                    hit[l] = hit[hd] && m >= 4; ml[l] = hit[l] ? m : 0; hback[l] = hback[hd];
                }
            }
            g_hist[nh]++;
            // greedy selection from the running anchor
            int covered[64];
            int pEnd = anchor;
            for (int l = 0; l < 64; l++) {
                int pos = w->p0 + l;
                covered[l] = pos < pEnd && pos > 0;     // strictly inside (pEnd is an END; the match start itself is never < its own end... start handled below)
                if (pos < pEnd) continue;
                if (!hit[l]) continue;
                int mstart = pos, mcand = (int)w->cand[l];
                int room = mstart - pEnd; if (mcand < room) room = mcand;
                int back = (l - myHead[l]) + (int)hback[l]; if (room < back) back = room;
                mstart -= back;
                out += seq_size(mstart - pEnd, pos + (int)ml[l] - mstart); g_seqs++;
                pEnd = pos + (int)ml[l];
            }
            // a match START is "visited" in the reference (it was probed and inserted): covered[] above marks
            // positions < pEnd at the time they are reached, which excludes starts.  Roll back the covered ones.
            if (!NOROLL)
                for (int l = 63; l >= 0; l--)
                    if (w->valid[l] && covered[l]) {
                        Ent *e = &table[w->h[l]];
                        if (e->used && e->pos == (uint16_t)(w->p0 + l)) *e = w->old[l];
                    }
            anchor = pEnd;
        }
        free(ring);
    }
    int last = n - anchor;
    out += 1 + last + (last >= 15 ? (last - 15) / 255 + 1 : 0);
    return (int)out;
}


// ------------------------------------------------------------------ pair form (round 3, ENC_PAIR)
// Two windows per step: both probed with every position written at once (W0 then W1), finished in order, covered
// positions take their insertion back (W1 first), the next pair starts at the end of the last selected match.
static int LATECONTIN = 0;
static int TABN = 4096;      // pair: table entries (the kernel: 4096; 3072 would fit 8 KiB of LDS with its tags -- round 6, five waves per SIMD)
static int BLINDBACK = 0;
static int SELFRUN = 0;       // pair: a position without a table candidate whose 4 bytes continue a run of the byte before it takes position - 1
static int DROP2ND = 0;       // pair: a window with more than 16 heads verifies its primary heads (left neighbour without a candidate) first and drops what does not fit 16 groups     // pair: covered positions put the old entry back without looking whether the bucket still holds them      // pair: W0's run-continuing lanes are not written at probe time, only (if uncovered) after the selection
static int sim_pair(const uint8_t *src, int n, int accel)
{
    static Ent table[4096];
    memset(table, 0, sizeof(table));
    long out = 0; int anchor = 0;
    if (n == 0) return 1;
    const uint32_t tmask = (1u << TAGBITS) - 1u;
    (void)accel;
    if (n >= 13) {
        const int mfl = n - 11, matchlimit = n - 5;
        int p0 = 0;
        while (p0 + 128 <= mfl) {
            int valid[128], candOk[128], hit[128], myHead[128], headOf[128], late[128]; uint32_t h[128], tg[128], cand[128], ml[128], hback[128]; Ent old[128];
            for (int w = 0; w < 2; w++) {
                for (int l = 64 * w; l < 64 * w + 64; l++) {
                    int pos = p0 + l; valid[l] = 1;
                    uint32_t hx = hash16(src + pos); h[l] = ((hx >> 8) * (uint32_t)TABN) >> 12; tg[l] = (hx >> (8 - TAGBITS)) & tmask;
                    old[l] = table[h[l]]; cand[l] = old[l].pos;
                    candOk[l] = old[l].used && cand[l] < (uint32_t)pos && (old[l].tag & tmask) == tg[l] && cand[l] >= 8;
                    if (SELFRUN && !candOk[l] && pos >= 9 && src[pos] == src[pos - 1] && src[pos + 1] == src[pos] && src[pos + 2] == src[pos] && src[pos + 3] == src[pos]) { cand[l] = pos - 1; candOk[l] = 1; }
                }
                for (int l = 64 * w; l < 64 * w + 64; l++) {
                    int contin = candOk[l] && l > 64 * w && candOk[l - 1] && cand[l] == cand[l - 1] + 1;
                    late[l] = LATECONTIN && w == 0 && contin;
                    if (!late[l]) { Ent e = { (uint16_t)(p0 + l), (uint8_t)tg[l], 1 }; table[h[l]] = e; }
                }
            }
            g_windows += 2;
            int hd = 0, nh[2] = {0, 0};
            int drop[128]; memset(drop, 0, sizeof drop);
            if (DROP2ND) for (int w = 0; w < 2; w++) {
                int tot = 0, prim = 0;
                for (int l = 64 * w; l < 64 * w + 64; l++) {
                    int contin = candOk[l] && l > 0 && candOk[l - 1] && cand[l] == cand[l - 1] + 1;
                    if (candOk[l] && !contin) { tot++; if (!(l > 0 && candOk[l - 1])) prim++; }
                }
                if (tot > 16) {
                    int room = 16 - prim;                       // groups left for secondary heads, in lane order
                    int pr = 0;
                    for (int l = 64 * w; l < 64 * w + 64; l++) {
                        int contin = candOk[l] && l > 0 && candOk[l - 1] && cand[l] == cand[l - 1] + 1;
                        if (!(candOk[l] && !contin)) continue;
                        int secondary = l > 0 && candOk[l - 1];
                        if (!secondary) { if (++pr > 16) drop[l] = 1; }
                        else if (room > 0) room--;
                        else drop[l] = 1;
                    }
                    g_ext2++;
                }
            }
            for (int l = 0; l < 128; l++) {
                hit[l] = 0; ml[l] = 0; hback[l] = 0; headOf[l] = 0;
                int contin = candOk[l] && l > 0 && candOk[l - 1] && cand[l] == cand[l - 1] + 1;
                int head = candOk[l] && !contin;
                int pos = p0 + l;
                if (head) {
                    hd = l; headOf[l] = 1; nh[l >> 6]++; g_heads++;
                    if (!drop[l] && nh[l >> 6] <= HEADCAP && rd32(src + pos) == rd32(src + cand[l])) {
                        hit[l] = 1;
                        int maxLen = matchlimit - pos;
                        int m = count_fwd(src, pos, cand[l], matchlimit);
                        ml[l] = m < maxLen ? m : maxLen;
                        int b = 0;
                        while (b < BACKCAP && b < (int)cand[l] && b < pos && src[pos - 1 - b] == src[cand[l] - 1 - b]) b++;
                        hback[l] = b;
                    }
                }
                myHead[l] = hd;
                if (contin) {
                    int m = (int)ml[hd] - (l - hd);
                    hit[l] = hit[hd] && m >= 4; ml[l] = hit[l] ? m : 0; hback[l] = hback[hd];
                }
            }
            int covered[128], pEnd = anchor, any = 0;
            for (int l = 0; l < 128; l++) {
                int pos = p0 + l;
                covered[l] = pos < pEnd;
                if (pos < pEnd || !hit[l]) continue;
                int mstart = pos, mcand = (int)cand[l];
                int room = mstart - pEnd; if (mcand < room) room = mcand;
                int back = (l - myHead[l]) + (int)hback[l]; if (room < back) back = room;
                mstart -= back;
                out += seq_size(mstart - pEnd, pos + (int)ml[l] - mstart); g_seqs++; any = 1;
                pEnd = pos + (int)ml[l];
            }
            for (int l = 127; l >= 0; l--) {
                Ent *e = &table[h[l]];
                if (late[l]) { if (!covered[l] && (!e->used || e->pos < (uint16_t)(p0 + l))) { Ent x = { (uint16_t)(p0 + l), (uint8_t)tg[l], 1 }; *e = x; } }
                else if (covered[l]) { if (BLINDBACK || (e->used && e->pos == (uint16_t)(p0 + l))) *e = old[l]; }
            }
            if (any) anchor = pEnd;
            p0 = p0 + 128 > pEnd ? p0 + 128 : pEnd;
        }
        // the rest of the block: the serial windows (as sim_cur), sharing the table
        // (approximation for the last < 128 positions: literals)
    }
    int last = n - anchor;
    out += 1 + last + (last >= 15 ? (last - 15) / 255 + 1 : 0);
    return (int)out;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "lzsynth";
    int bl = argc > 2 ? atoi(argv[2]) : 65536, nb = argc > 3 ? atoi(argv[3]) : 64;
    uint8_t *buf = malloc(bl), *dst = malloc(bl + bl / 255 + 64);
    long ref = 0;
    uint8_t **blocks = malloc(sizeof(uint8_t *) * nb);
    for (int b = 0; b < nb; b++) {
        blocks[b] = malloc(bl);
        if (!strcmp(kind, "lzsynth")) orc_gen_lzsynth(blocks[b], bl, b, 16, 2048);
        else if (!strcmp(kind, "text")) orc_gen_text(blocks[b], bl, b);
        else if (!strcmp(kind, "zeros")) memset(blocks[b], 0, bl);
        else if (!strcmp(kind, "file")) { FILE *f = fopen(argv[4], "rb"); fseek(f, (long)b * bl, SEEK_SET); size_t got = fread(blocks[b], 1, bl, f); if ((int)got < bl) memset(blocks[b] + got, 0, bl - got); fclose(f); }
        else orc_gen_random(blocks[b], bl, b);
        ref += orc_compress_block(blocks[b], dst, bl, bl + bl / 255 + 16, 1);
    }
    printf("%s %d x %d: reference ratio %.4f\n", kind, nb, bl, (double)nb * bl / ref);
#define RUN(label, call) do { long s = 0; g_windows = g_heads = g_seqs = g_hits = g_ext2 = 0; for (int b = 0; b < nb; b++) s += call; \
    printf("  %-34s ratio %.4f  windows/blk %.0f heads/win %.1f hits/win %.2f seqs/blk %.0f long(>56)/win %.2f\n", label, (double)nb * bl / s, \
        (double)g_windows / nb, (double)g_heads / (g_windows ? g_windows : 1), (double)g_hits / (g_windows ? g_windows : 1), (double)g_seqs / nb, (double)g_ext2 / (g_windows ? g_windows : 1)); } while (0)
    TAGBITS = 4; RUN("cur tag4", sim_cur(blocks[b], bl, 1));
    TAGBITS = 4; END2 = 1; RUN("cur tag4 +end2", sim_cur(blocks[b], bl, 1)); END2 = 0;
    TAGBITS = 4; END2 = 1; END2MIN = 24; RUN("cur tag4 +end2 (len>=24)", sim_cur(blocks[b], bl, 1)); END2MIN = 40; RUN("cur tag4 +end2 (len>=40)", sim_cur(blocks[b], bl, 1)); END2MIN = 20; RUN("cur tag4 +end2 (len>=20)", sim_cur(blocks[b], bl, 1)); END2MIN = 16; RUN("cur tag4 +end2 (len>=16)", sim_cur(blocks[b], bl, 1)); END2MIN = 12; RUN("cur tag4 +end2 (len>=12)", sim_cur(blocks[b], bl, 1)); END2 = 0; END2MIN = 0;
    TAGBITS = 4; MINC = 8; RUN("cur tag4 cand>=8", sim_cur(blocks[b], bl, 1)); MINC = 0;
    TAGBITS = 4; BACKCAP = 0; RUN("cur tag4 back0", sim_cur(blocks[b], bl, 1)); BACKCAP = 16; RUN("cur tag4 back16", sim_cur(blocks[b], bl, 1)); BACKCAP = 1000; RUN("cur tag4 back-inf", sim_cur(blocks[b], bl, 1)); BACKCAP = 8;
    TAGBITS = 4; WIN = 128; RUN("cur tag4 win128", sim_cur(blocks[b], bl, 1)); WIN = 64;
    TAGBITS = 4; HEADCAP = 32; RUN("cur tag4 headcap32", sim_cur(blocks[b], bl, 1)); HEADCAP = 48; RUN("cur tag4 headcap48", sim_cur(blocks[b], bl, 1)); HEADCAP = 32; RUN("pair tag4", sim_pair(blocks[b], bl, 1)); HEADCAP = 48; RUN("pair tag4 headcap48", sim_pair(blocks[b], bl, 1)); HEADCAP = 1000; RUN("pair tag4 no head cap", sim_pair(blocks[b], bl, 1)); HEADCAP = 32; LATECONTIN = 1; RUN("pair tag4 late contin", sim_pair(blocks[b], bl, 1)); LATECONTIN = 0; BLINDBACK = 1; RUN("pair tag4 blind takeback", sim_pair(blocks[b], bl, 1)); BLINDBACK = 0; HEADCAP = 1000; SELFRUN = 1; RUN("pair tag4 self runs (offset 1)", sim_pair(blocks[b], bl, 1)); TABN = 3072; RUN("pair tag4 self runs, 3072 entries", sim_pair(blocks[b], bl, 1)); TABN = 2048; RUN("pair tag4 self runs, 2048 entries", sim_pair(blocks[b], bl, 1)); TABN = 4096; TAGBITS = 0; RUN("pair no tags self runs, 4096 entries", sim_pair(blocks[b], bl, 1)); TAGBITS = 4; SELFRUN = 0; DROP2ND = 1; RUN("pair tag4 one round, primary heads first", sim_pair(blocks[b], bl, 1)); DROP2ND = 0; HEADCAP = 16; RUN("pair tag4 headcap16", sim_pair(blocks[b], bl, 1)); HEADCAP = 32; HEADCAP = 1000;
    TAGBITS = 8; RUN("cur tag8", sim_cur(blocks[b], bl, 1));
    TAGBITS = 0; RUN("cur tag0", sim_cur(blocks[b], bl, 1));
    for (int tb = 4; tb <= 8; tb += 4)
        for (int d = 1; d <= 3; d++) {
            char lab[64]; TAGBITS = tb; DEPTH = d; NOROLL = 0;
            snprintf(lab, sizeof lab, "pipe tag%d depth%d", tb, d); RUN(lab, sim_pipe(blocks[b], bl, 1));
        }
    TAGBITS = 8; DEPTH = 2; NOROLL = 1; RUN("pipe tag8 insert-all", sim_pipe(blocks[b], bl, 1));
    TAGBITS = 8; DEPTH = 2; NOROLL = 0; KNOWN = 0; RUN("pipe tag8 depth2 noknown", sim_pipe(blocks[b], bl, 1)); KNOWN = 1;
    memset(g_hist, 0, sizeof g_hist);
    RUN("pipe tag8 depth2", sim_pipe(blocks[b], bl, 1));
    { long tot = 0, c16 = 0, c32 = 0, c8 = 0; for (int i = 0; i <= 64; i++) { tot += g_hist[i]; if (i > 16) c16 += g_hist[i]; if (i > 32) c32 += g_hist[i]; if (i > 8) c8 += g_hist[i]; }
      printf("  heads/window: P(>8) %.3f P(>16) %.3f P(>32) %.3f\n", (double)c8 / tot, (double)c16 / tot, (double)c32 / tot); }
    (void)buf;
    return 0;
}
