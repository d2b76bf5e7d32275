// decode_par2.hpp -- the lane-parallel decoder of decode_par.hpp split over TWO wavefronts per
// block: a PARSER wave and a COPIER wave that work on consecutive batches at the same time.
//
//   parser  (wave 0): window -> speculative parse -> pointer-jumping chain -> per-lane sequence
//                     fields, relative output positions (DPP scan) and dependency masks; it writes
//                     one batch descriptor into LDS and immediately goes on to the NEXT window,
//                     speculating that the copier will accept the whole batch (it almost always does).
//   copier  (wave 1): takes a descriptor, applies the output-side conditions, copies far matches,
//                     literals and near matches into the LDS ring, flushes the ring to memory.  It owns
//                     the authoritative (ip, op); when it accepted fewer sequences than the parser
//                     assumed (or a sequence needed the sequential decoder), the parser's speculative
//                     batch is dropped and re-parsed from the authoritative position.
//
// One s_barrier per batch keeps the two in lock step (descriptor and window are double buffered).
// The point is occupancy per LDS byte and per register: each wave carries about half the state and
// half the instructions of the single-wave decoder, so a CU holds ~50 % more waves per block in
// flight.  Results are bit-identical to decode_par.hpp / decode_seq.hpp by construction: the same
// "plain interior sequence" rules decide what is handled here, everything else goes to
// decode_seq_run() (on the copier wave), which also produces the reference's exact error codes.
#pragma once

#include "decode_par.hpp"

namespace lz4dev {

#ifndef P2_RING
#define P2_RING 6144
#endif
#ifndef P2_HIST
#define P2_HIST 2048
#endif
#ifndef P2_BATCH_OUT
#define P2_BATCH_OUT 2048
#endif
#ifndef P2_WAVES
#define P2_WAVES 6
#endif

struct __attribute__((aligned(16))) Par2Lds {
    uint8_t win[2][PAR_WIN + 32];
    uint16_t jump[PAR_NODES + 8];
    uint64_t dA[2][LZ4_WAVE];     // litStart | lit<<16 | ml<<32 | off16<<48
    uint64_t dB[2][LZ4_WAVE];     // relStart | nxt<<32
    uint64_t dN[2][LZ4_WAVE];     // dependency mask
    int hdr[2][4];                // nseqSpec, ipStart, ipW0, ipNextSpec
    int ctl[4];                   // authoritative ip after the copier's last round, done, result
    uint8_t ring[P2_RING + 32];
};

// ------------------------------------------------------------------------------------------------
// parser wave: parse one batch that starts at block position `ip` into slot `s`
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void par2_parse(Par2Lds &L, int s, int ip, const uint8_t *src, int iend,
                                           const uint8_t *bufLo, const uint8_t *bufHi)
{
    const int lane = lane_id();
    const uint8_t *jumpB = (const uint8_t *)L.jump;
    uint8_t *win = L.win[s];

    // ---- window ----
    const uint8_t *gp = src + ip;
    const uintptr_t abase = (uintptr_t)gp & ~(uintptr_t)15;
    const int wofs = (int)((uintptr_t)gp - abase);
    const int ipW0 = ip - wofs;
    const int iendW = iend - ipW0;
    const int inLim = min(iendW - 32, PAR_WIN);
    {
        const uint8_t *q = (const uint8_t *)(abase + 16u * (uint32_t)lane);
        uint4 v;
        if (q >= bufLo && q + 16 <= bufHi) {
            v = *(const uint4 *)q;
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++)
                if (q + k >= bufLo && q + k < bufHi) w[k >> 2] |= (uint32_t)q[k] << (8 * (k & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *(uint4 *)&win[16 * lane] = v;
    }
    wave_fence();

    // ---- speculative parse (registers only) ----
    uint32_t J[8];
    {
        const uint64_t lo = *(const uint64_t *)&win[8 * lane];
        const uint32_t hi = (uint32_t)win[8 * lane + 8];
        const uint32_t nodeLim = (uint32_t)min(inLim, PAR_NODES - 1);
        const uint32_t base3 = 8u * (uint32_t)lane + 3u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t t = (uint32_t)(lo >> (8 * j)) & 0xffu;
            const uint32_t b1 = (j < 7) ? ((uint32_t)(lo >> (8 * (j + 1))) & 0xffu) : hi;
            const uint32_t lit0 = t >> 4;
            uint32_t nxt = base3 + (uint32_t)j + ((lit0 == 15u) ? 16u + b1 : lit0);
            nxt += ((t & 15u) == 15u) ? 1u : 0u;
            J[j] = ((int)nxt <= (int)nodeLim) ? 2u * nxt : (uint32_t)PAR_END;
        }
        *(uint4 *)&L.jump[8 * lane] =
            make_uint4(J[0] | (J[1] << 16), J[2] | (J[3] << 16), J[4] | (J[5] << 16), J[6] | (J[7] << 16));
    }
    wave_fence();

    // ---- chain: sequence r -> lane r ----
    uint32_t c2 = (lane == 0) ? 2u * (uint32_t)wofs : (uint32_t)PAR_END;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int d = 1 << k;
        const int cj = (int)*(const uint16_t *)(jumpB + c2);
        if (k < 5) {
#pragma unroll
            for (int j = 0; j < 8; j++) J[j] = (uint32_t)*(const uint16_t *)(jumpB + J[j]);
        }
        int sh;
        if (k == 0) sh = par_row_shr<1>(cj);
        else if (k == 1) sh = par_row_shr<2>(cj);
        else if (k == 2) sh = par_row_shr<4>(cj);
        else if (k == 3) sh = par_row_shr<8>(cj);
        else sh = par_bperm(cj, (lane - d) & 63);
        if (lane >= d && lane < 2 * d) c2 = (uint32_t)sh;
        if (k < 5) {
            wave_fence();
            *(uint4 *)&L.jump[8 * lane] =
                make_uint4(J[0] | (J[1] << 16), J[2] | (J[3] << 16), J[4] | (J[5] << 16), J[6] | (J[7] << 16));
            wave_fence();
        }
    }

    // ---- own sequence ----
    const bool has = c2 < (uint32_t)PAR_END;
    const uint32_t cc = has ? (c2 >> 1) : 0u;
    const uint32_t tb = (uint32_t)(*(const par_u16u *)&win[cc]);
    const uint32_t t = tb & 0xffu, b1 = tb >> 8;
    const bool is15 = (t >> 4) == 15u;
    const uint32_t lit = is15 ? 15u + b1 : (t >> 4);
    const uint32_t litStart = cc + 1u + (is15 ? 1u : 0u);
    const uint32_t offPos = litStart + lit;
    const uint32_t ob = *(const par_u32u *)&win[offPos];
    const uint32_t off16 = ob & 0xffffu, b2 = (ob >> 16) & 0xffu;
    const bool mlx = (t & 15u) == 15u;
    const uint32_t ml = (t & 15u) + LZ4_MINMATCH + (mlx ? b2 : 0u);
    const uint32_t nxt = offPos + 2u + (mlx ? 1u : 0u);
    bool ok = has && !(is15 && b1 == 255u) && !(mlx && b2 == 255u) && (int)nxt <= inLim && off16 != 0;
    int len = ok ? (int)(lit + ml) : 0;
    int incl = par_scan_incl(len);
    ok = ok && incl <= P2_BATCH_OUT;
    const uint64_t okm = __ballot(ok);
    const int nseq = (~okm) ? (int)__builtin_ctzll(~okm) : LZ4_WAVE;
    const bool act = lane < nseq;
    const int rel = incl - len;                       // output position relative to the batch start
    const int sposRel = rel + (int)lit - (int)off16;  // match source, relative (negative: before the batch)

    // ---- dependency mask: sequences of this batch my source [spos, min(spos+ml, myStart)) overlaps ----
    uint64_t need = 0;
    {
        const int srcHi = min(sposRel + (int)ml, rel);
        const int xlo = max(sposRel, 0), xhi = max(srcHi - 1, 0);
        int jlo = 0, jhi = 0;
#pragma unroll
        for (int stp = 32; stp >= 1; stp >>= 1) {
            const int c1 = jlo + stp, cb = jhi + stp;
            const int v1 = par_bperm(rel, c1 & 63), v2 = par_bperm(rel, cb & 63);
            if (c1 < nseq && v1 <= xlo) jlo = c1;
            if (cb < nseq && v2 <= xhi) jhi = cb;
        }
        if (act && srcHi > 0 && srcHi > sposRel) {
            const uint64_t upto = (jhi >= 63) ? ~0ull : ((1ull << (jhi + 1)) - 1ull);
            need = upto & ~((1ull << jlo) - 1ull);
            need &= ~(1ull << lane);
        }
    }

    // ---- descriptor ----
    L.dA[s][lane] = (uint64_t)litStart | ((uint64_t)lit << 16) | ((uint64_t)ml << 32) | ((uint64_t)off16 << 48);
    L.dB[s][lane] = (uint64_t)(uint32_t)rel | ((uint64_t)nxt << 32);
    L.dN[s][lane] = need;
    if (lane == 0) {
        L.hdr[s][0] = nseq;
        L.hdr[s][1] = ip;
        L.hdr[s][2] = ipW0;
    }
    const int nxtLast = __builtin_amdgcn_readlane((int)nxt, (nseq > 0 ? nseq : 1) - 1);
    if (lane == 0) L.hdr[s][3] = (nseq > 0) ? ipW0 + nxtLast : -1;
    wave_fence();
}

// ------------------------------------------------------------------------------------------------
// the two-wave block decoder.  Called by BOTH waves of a 128-thread workgroup with the same
// (wave-uniform) arguments; `role` 0 = parser, 1 = copier.  Returns the block result on the copier.
// ------------------------------------------------------------------------------------------------
__device__ int decode_block_par2(int role, const uint8_t *src, int srcLen, uint8_t *dst, int cap, const uint8_t *bufLo,
                                 const uint8_t *bufHi, Par2Lds &L)
{
    const int lane = lane_id();
    const int iend = srcLen;

    if (role == 0) {
        // ======================= parser =======================
        if (lane == 0) L.jump[PAR_NODES] = PAR_END;
        par2_parse(L, 0, 0, src, iend, bufLo, bufHi);
        __syncthreads();
        for (int r = 0;; r++) {
            const int s = r & 1;
            const int authIp = uni(L.ctl[0]);
            if (uni(L.ctl[1])) break;
            const bool slotOk = uni(L.hdr[s][1]) == authIp;
            const int nextIp = slotOk ? uni(L.hdr[s][3]) : authIp;
            if (nextIp >= 0 && iend - nextIp >= 64) {
                par2_parse(L, s ^ 1, nextIp, src, iend, bufLo, bufHi);
            } else if (lane == 0) {
                L.hdr[s ^ 1][1] = -1;                                  // nothing parsed: the copier decides
            }
            __syncthreads();
        }
        return 0;
    }

    // ======================= copier =======================
    const uint32_t A = (uint32_t)((uintptr_t)dst & 15);
    int ip = 0, op = 0;
    int ringBase = 0, flushed = 0;
    SeqState st;

    auto flush = [&](int upto, bool final) {
        wave_fence();
        int f = flushed;
        const int mis = (int)((A + (uint32_t)f) & 15u);
        if (mis) {
            const int head = 16 - mis;
            if (upto - f >= head) {
                if (lane < head) dst[f + lane] = L.ring[f - ringBase + (int)A + lane];
                f += head;
            } else if (!final) {
                return;
            }
        }
        if (((A + (uint32_t)f) & 15u) == 0) {
            const int n16 = (upto - f) >> 4;
            for (int c = lane; c < n16; c += LZ4_WAVE) {
                const uint4 v = *(const uint4 *)&L.ring[f - ringBase + (int)A + 16 * c];
                *(uint4 *)(dst + f + 16 * c) = v;
            }
            f += n16 << 4;
        }
        if (final) {
            for (int x = f + lane; x < upto; x += LZ4_WAVE) dst[x] = L.ring[x - ringBase + (int)A];
            f = upto;
        }
        flushed = f;
        wave_fence();
    };

    int result = 0;
    bool done = false;
    if (lane == 0) { L.ctl[0] = 0; L.ctl[1] = 0; }
    __syncthreads();
    for (int r = 0;; r++) {
        const int s = r & 1;
        const bool tail = (iend - ip < 64 || cap - op < 128);
        const bool slotOk = !tail && uni(L.hdr[s][1]) == ip;
        bool slow = tail;                                   // sequential decoder needed this round?
        if (slotOk) {
            const uint8_t *win = L.win[s];
            const int nseqSpec = uni(L.hdr[s][0]);
            const int ipW0 = uni(L.hdr[s][2]);
            const uint64_t a = L.dA[s][lane], b = L.dB[s][lane];
            const uint32_t litStart = (uint32_t)a & 0xffffu, lit = (uint32_t)(a >> 16) & 0xffffu;
            const uint32_t ml = (uint32_t)(a >> 32) & 0xffffu, off16 = (uint32_t)(a >> 48);
            const int rel = (int)(uint32_t)b;
            const uint32_t nxt = (uint32_t)(b >> 32);
            const int outStart = op + rel;
            const int outEnd = outStart + (int)(lit + ml);
            const int dpos = outStart + (int)lit;
            const int spos = dpos - (int)off16;
            const bool ok = lane < nseqSpec && outEnd + 64 < cap && spos >= 0 &&
                            (spos >= ringBase || spos + (int)ml <= flushed);
            const uint64_t okm = __ballot(ok);
            const int nseq = (~okm) ? (int)__builtin_ctzll(~okm) : LZ4_WAVE;
            if (nseq == 0) {
                slow = true;
            } else {
                const bool act = lane < nseq;
                const int opNext = __builtin_amdgcn_readlane(outEnd, nseq - 1);
                const int ipNext = ipW0 + __builtin_amdgcn_readlane((int)nxt, nseq - 1);
                const uint32_t mdA = (uint32_t)(dpos - ringBase) + A;
                const bool nearSrc = spos >= ringBase;
                const bool w8 = ml >= 8 && off16 >= 8;
                const bool w4 = !w8 && off16 >= 4;
                const bool grp = w8 && (off16 >= 32 || off16 >= ml);

                // ---- far matches: source already in global memory ----
                const uint64_t farm = __ballot(act && !nearSrc);
                if (farm) {
                    const uint8_t *gsrc = dst + spos;
                    const bool mine = act && !nearSrc;
                    const uint32_t step = (ml >= 8) ? 8u : 4u;
                    const uint32_t last = ml - step;
                    for (uint32_t base = 0; __ballot(mine && base < ml); base += 32) {
                        if (mine && base < ml) {
                            if (step == 8) {
                                uint64_t v[4];
                                uint32_t o[4];
#pragma unroll
                                for (int k = 0; k < 4; k++) { o[k] = min(base + 8u * k, last); v[k] = *(const par_u64u *)(gsrc + o[k]); }
#pragma unroll
                                for (int k = 0; k < 4; k++) *(par_u64u *)&L.ring[mdA + o[k]] = v[k];
                            } else {
                                const uint32_t v0 = *(const par_u32u *)(gsrc);
                                const uint32_t v1 = *(const par_u32u *)(gsrc + last);
                                *(par_u32u *)&L.ring[mdA] = v0;
                                *(par_u32u *)&L.ring[mdA + last] = v1;
                            }
                        }
                    }
                }

                // ---- literals: window -> ring ----
                {
                    const uint32_t sA = litStart;
                    const uint32_t dA = (uint32_t)(outStart - ringBase) + A;
                    const uint32_t n = act ? lit : 0u;
                    if (n >= 8) {
                        const uint32_t last = n - 8;
                        for (uint32_t o = 0;; o += 8) {
                            const uint32_t oo = min(o, last);
                            *(par_u64u *)&L.ring[dA + oo] = *(const par_u64u *)&win[sA + oo];
                            if (o >= last) break;
                        }
                    } else if (n >= 4) {
                        const uint32_t v0 = *(const par_u32u *)&win[sA], v1 = *(const par_u32u *)&win[sA + n - 4];
                        *(par_u32u *)&L.ring[dA] = v0;
                        *(par_u32u *)&L.ring[dA + n - 4] = v1;
                    } else if (n > 0) {
                        const uint32_t v = *(const par_u32u *)&win[sA];
                        L.ring[dA] = (uint8_t)v;
                        if (n > 1) L.ring[dA + 1] = (uint8_t)(v >> 8);
                        if (n > 2) L.ring[dA + 2] = (uint8_t)(v >> 16);
                    }
                }
                wave_fence();

                // ---- near matches: dependency rounds ----
                uint64_t need = (act && nearSrc) ? (L.dN[s][lane] & ~farm) : 0ull;
                uint64_t doneM = ((nseq >= LZ4_WAVE) ? 0ull : (~0ull << nseq)) | farm;
                bool pending = act && nearSrc;
                const uint32_t msA = nearSrc ? (uint32_t)(spos - ringBase) + A : 0u;
                const uint32_t last8 = ml - 8;
                while (~doneM) {
                    const bool mine = pending && ((need & ~doneM) == 0ull);
                    for (uint32_t base = 0; __ballot(mine && grp && base < ml); base += 32) {
                        if (mine && grp && base < ml) {
                            uint64_t v[4];
                            uint32_t o[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) { o[k] = min(base + 8u * k, last8); v[k] = *(const par_u64u *)&L.ring[msA + o[k]]; }
#pragma unroll
                            for (int k = 0; k < 4; k++) *(par_u64u *)&L.ring[mdA + o[k]] = v[k];
                        }
                        wave_fence();
                    }
                    if (__ballot(mine && !grp)) {
                        const uint32_t step = w8 ? 8u : (w4 ? 4u : 1u);
                        const uint32_t last = ml - step;
                        const bool slowCopy = mine && !grp;
                        for (uint32_t o = 0; __ballot(slowCopy && o < ml); o += step) {
                            if (slowCopy && o < ml) {
                                const uint32_t oo = min(o, last);
                                if (w8) *(par_u64u *)&L.ring[mdA + oo] = *(const par_u64u *)&L.ring[msA + oo];
                                else if (w4) *(par_u32u *)&L.ring[mdA + oo] = *(const par_u32u *)&L.ring[msA + oo];
                                else L.ring[mdA + oo] = L.ring[msA + oo];
                            }
                            wave_fence();
                        }
                    }
                    pending = pending && !mine;
                    doneM |= __ballot(mine);
                }
                wave_fence();

                // ---- advance, flush, slide ----
                op = opNext;
                ip = ipNext;
                flush(op, false);
                if (op - ringBase + (int)A + P2_BATCH_OUT + 32 > P2_RING) {
                    const int newBase = (op - P2_HIST) & ~15;
                    const int delta = newBase - ringBase;
                    const int n16 = (op - newBase + (int)A + 15) >> 4;
                    for (int k = lane; k < n16; k += LZ4_WAVE) {
                        const uint4 v = *(const uint4 *)&L.ring[delta + 16 * k];
                        wave_fence();
                        *(uint4 *)&L.ring[16 * k] = v;
                    }
                    ringBase = newBase;
                    wave_fence();
                }
            }
        }
        if (slow) {
            // one sequence through the sequential decoder (or the whole tail of the block)
            flush(op, true);
            st.ip = ip; st.op = op; st.fast = true;
            int rr = decode_seq_run(st, tail ? 0 : 1, src, srcLen, dst, cap, nullptr, 0, bufLo, bufHi);
            if (rr == SEQ_CONTINUE && (!st.fast || iend - st.ip < 64 || cap - st.op < 128))
                rr = decode_seq_run(st, 0, src, srcLen, dst, cap, nullptr, 0, bufLo, bufHi);
            if (rr != SEQ_CONTINUE) {
                result = rr;
                done = true;
            } else {
                ip = st.ip; op = st.op;
                wave_fence();
                ringBase = (op > P2_HIST) ? ((op - P2_HIST) & ~15) : 0;
                flushed = op;
                for (int x = ringBase + lane; x < op; x += LZ4_WAVE) L.ring[x - ringBase + (int)A] = dst[x];
                wave_fence();
            }
        }
        if (lane == 0) { L.ctl[0] = ip; L.ctl[1] = done ? 1 : 0; }
        __syncthreads();
        if (done) break;
    }
    return result;
}

} // namespace lz4dev
