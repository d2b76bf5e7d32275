// kernels.h -- launch interface between the C-ABI layer (api.cpp) and kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// per-block header rejection codes (mirrors MI355LZ4_BLK_E_* in include/mi355lz4.h)
#define BLK_E_COMPLEN   (-0x7F000001)
#define BLK_E_TRUNCATED (-0x7F000002)
#define BLK_E_UNCOMPLEN (-0x7F000003)
// LZ4_compressBound(LZ4_MAX_INPUT_SIZE): reference lz4_MAX_OUTPUT_SIZE, Internal/LZ4.hs:145-147
#define MAX_COMP_LEN 2122219150

struct DecodeArgs {
    const uint8_t *framed;
    uint64_t framedLen;
    const uint64_t *blockOff;
    int nBlocks;
    int headerKind;
    int fixedUncomp;
    int linked;
    uint8_t *out;
    const uint64_t *outOff;
    const int32_t *outCap;   // may be null
    int32_t *result;
    const uint8_t *dict0;    // linked only: dictionary in force before block 0 (may be null)
    uint32_t dict0Len;
    const int32_t *streamFirst;   // linked only: stream s = blocks [streamFirst[s], streamFirst[s+1]); null = one stream
    int nStreams;
    int lookBack;                 // linked, one stream: blocks of the SAME stream that precede block 0 in result[] /
                                  // outOff[] (already final); lets a long stream be decoded range by range
    // deferred-copy decode of one long linked stream (linked_replay.hpp): per-block state of the tolerant pass
    // and the pool its deferred lists live in; all null when the pool is not available
    void *tolPool;                // TolEntry[tolRegions][TOL_LIST_CAP]
    int tolRegions;
    int tolPer;                   // list regions per block of a tolerant launch: block segFirst + i owns regions [i * tolPer, (i + 1) * tolPer)
    uint32_t *tolCounter;         // regions handed out so far
    int32_t *tolRegion;           // per block: region index, or -1 (no list: serial path)
    int32_t *tolCount;            // per block: entries appended (may exceed the capacity: overflow)
    int32_t *tolSize;             // per block: result of the tolerant decode
    // one long linked stream, data-parallel second pass (linked_ptr.hpp)
    uint32_t *linkStat;           // filled by the standalone pass: {blocks with a codec error, first, last}
    int segFirst, segEnd;         // blocks the second-pass kernels cover in this launch
    uint32_t *ptr;                // source pointers of the segment's bytes
    uint64_t ptrCap;              // ... capacity in pointers
    void *ptrCtl;                 // PtrCtl
    uint32_t *ptrBad;             // per stream (one entry without streamFirst): left to the serial walk
    int asyncGate;                // second-pass kernels return at once when linkStat[0] == 0 (asynchronous linked decode)
    int onlyBlk;                  // >= 0: the fetch covers this block alone (mi355lz4_decompress_linked_end_last); -1: all
    // run starts of a linked stream's dependent blocks (k_run_starts -> k_decode_fixup_runs): runList[0] = count,
    // runList[1..] = first block of each run, at most runCap of them; null: the launch covers one run that starts at segFirst
    int32_t *runList;
    int runCap;
    // long linked streams, run-in decode (kernels.hip, k_runin_*): pieces of runPiece consecutive blocks of [segFirst, segEnd)
    uint8_t *ring;                // scratch: two blocks per piece (piece p at ring + p * 2 * ringStride)
    uint64_t ringStride;          // >= the largest block capacity
    const uint8_t *zeroPage;      // 64 KiB of 0x00: the stand-in for the dictionary of the block a run-in starts at
    int runPiece;                 // blocks per piece
    int runIn;                    // blocks a piece decodes in front of its own (<= 64)
    int runSpin;                  // k_runin_fix: how many times a piece polls for the piece in front of it before it leaves itself to the next round
    int runRound;                 // k_runin_fix: the round this launch is
    int32_t *runRes;              // per block of the segment: its result (result[] is written when everything is final)
    int32_t *runInfo;             // per piece {block the run-in's dictionary came from | -1 | -2 exact, its length, its ring slot, -}
    uint32_t *runDirty;           // per piece: the round in which it is to be redone, 0xffffffff = none
    uint32_t *runCtl;             // [0] pieces marked for the next round, [1] give the call up (1: a block failed, 2: a chain of dirty pieces)
    // experiment builds only (MI355LZ4_EXPERIMENTS; decode_par.hpp, LIST): block blk's token list lives at tokList +
    // blockOff[blk] / 2 (a sequence is at least three compressed bytes), tokCnt[blk] entries; both null without the list pass
    uint8_t *tokList;
    int32_t *tokCnt;
    // diagnostics of the workgroup-per-block decoder (mi355lz4_debug_cu): 16 words per block, null = off
    uint32_t *cuDbg;
    int cuBail;                 // workgroup-per-block decoder: leave blocks that do not suit it (hardly compressible; long literal runs) to the lane-parallel one at once (decoder variant 0; variant 4 keeps them)
    // big linked blocks by the workgroup-per-block decoder (launch_cu_linked): every dependent block decoded with a guess of its
    // dictionary -- zeros in pass 1, from pass 2 on a snapshot of the last 64 KiB its predecessor decoded to in the pass before --
    // until no snapshot changes any more
    uint8_t *cuSnap;            // [nBlocks][65536]
    uint32_t *cuFlags;          // [0] snapshots that changed in the last launch_cu_tails, [1] blocks the form cannot take, [2 + k] block k's snapshot changed
    int32_t *cuRes;             // [nBlocks] results of the passes (published by the caller when the snapshots have settled)
    int cuPass;
};

struct EncodeArgs {
    const uint8_t *src;
    const uint64_t *srcOff;  // may be null -> blk * blockStride
    const int32_t *srcLen;   // may be null -> uniformLen
    uint64_t blockStride;
    int uniformLen;
    int nBlocks;
    int accel;
    int headerKind;
    uint8_t *slots;
    size_t slotStride;
    int32_t *framedLen;
    unsigned long long *stats;   // diagnostics only (ENC_STATS builds); may be null
    int linked;                  // the blocks are consecutive blocks of ONE stream: block i-1 is block i's dictionary
    int lookBack;                // linked: blocks of the same stream that precede block 0 in srcOff[] / srcLen[]
};

// small batches: a block's segments are compressed by several waves (kernels.hip, K2 small batches)
struct EncodeSegArgs {
    EncodeArgs e;
    int segs;                    // segments per block
    int segLen;                  // bytes per segment (the last one takes the rest)
    uint64_t *lists;             // sequence records: block b's segment j at lists + b * listStride + s0 / 4 + j
    size_t listStride;           // records per block: maxBlockLen / 4 + segs + 1
    uint32_t *segCount;          // records per segment
    uint32_t *segBytes;          // bytes a segment's records emit to (k_seg_sizes)
    int32_t *segPrevEnd;         // end of the last sequence in front of the segment
};
void launch_encode_seg(const EncodeSegArgs &a, hipStream_t s);
void launch_decode_seq(const DecodeArgs &a, hipStream_t s);
void launch_decode_par(const DecodeArgs &a, unsigned long long *stats, hipStream_t s);
void launch_decode_cu(const DecodeArgs &a, hipStream_t s);       // one workgroup per block (decode_cu.hpp): calls that do not fill the GPU
#ifdef MI355LZ4_EXPERIMENTS
void launch_decode_tok(const DecodeArgs &a, hipStream_t s);      // token lists (a.tokList / a.tokCnt), then the list-driven decoder
#endif
#define PAR_STATS_COUNT 32
void launch_linked_tolerant(const DecodeArgs &a, hipStream_t s);   // both cover blocks [a.segFirst, a.segEnd)
void launch_linked_resolve(const DecodeArgs &a, hipStream_t s);
void launch_linked_resolve_a(const DecodeArgs &a, hipStream_t s);   // pointers only (no output byte is read)
void launch_linked_resolve_b(const DecodeArgs &a, hipStream_t s);   // data: fetch, finish, fallbacks
void launch_linked_fetch_block(const DecodeArgs &a, hipStream_t s);  // data: the fetch of block a.onlyBlk alone; PtrCtl::lastOpen tells whether it is complete
size_t ptr_ctl_last_open_offset();
void launch_longest_stream(const DecodeArgs &a, hipStream_t s);   // linkStat[3]
void launch_cu_linked(const DecodeArgs &a, bool decode, hipStream_t s);   // one pass over the dependent blocks (a.cuPass; decode = false: the first launch has made it), then the snapshots
void launch_cu_publish(const DecodeArgs &a, hipStream_t s);      // a.cuRes -> a.result for the dependent blocks
void launch_dict_share(const DecodeArgs &a, int step, int count, hipStream_t s);   // linkStat[8], [9]: bytes taken directly from the dictionary / bytes walked, over `count` blocks from a.segFirst on
void launch_link_stat(const DecodeArgs &a, hipStream_t s);       // linkStat from result[] (the decode launchers call it themselves)
void launch_runin_decode(const DecodeArgs &a, hipStream_t s);    // long linked stream: every piece with its run-in + the comparison
void launch_runin_fix(const DecodeArgs &a, hipStream_t s);       // ... one round (a.runRound) of pieces to be redone
void launch_runin_publish(const DecodeArgs &a, hipStream_t s);   // ... results into result[]
void launch_linked_runs(const DecodeArgs &a, hipStream_t s);      // one stream, short runs of dependent blocks: one wave per run, exact decoder with dictionary
size_t ptr_ctl_bytes();
size_t tol_region_bytes();
void launch_encode(const EncodeArgs &a, bool bigBlocks, hipStream_t s);   // bigBlocks: some block is above 64 KiB
void launch_compact(const uint8_t *slots, size_t slotStride, const int32_t *framedLen, int nBlocks,
                    uint8_t *dense, size_t denseCap, uint64_t *denseOff, hipStream_t s);
void launch_interleave(const uint8_t *local, const uint64_t *localOff, int nLocal, int rank, int nRanks,
                       uint8_t *global, const uint64_t *globalOff, hipStream_t s);
void launch_index(const uint8_t *framed, uint64_t framedLen, const uint64_t *blockOff, int nBlocks,
                  int headerKind, int fixedUncomp, int32_t *scratchSizes, uint64_t *outOff, hipStream_t s);
void launch_generate(int kind, uint8_t *dst, int blockLen, int nBlocks, uint64_t firstBlock,
                     uint64_t blockStep, uint32_t litMax, uint32_t offMax, hipStream_t s);
