// linked_ptr.hpp -- second pass of the deferred-copy decode of ONE long linked stream, without the serial walk.
//
// The tolerant pass (decode_par.hpp / decode_seq.hpp, TolCtx) has decoded every dependent block in parallel and
// left, per block, the list of matches it could not copy: those that start in the previous block's output
// (cbits/lz4.c:1883-1911, the ext-dict case of LZ4_decompress_safe_continue, :2347-2355) and those that read
// bytes such a match produces.  What is still missing is data, not structure: WHERE every byte of the stream
// comes from is known, all the way back, because a block's size and its match offsets depend on its tokens
// only.  So the chain of blocks is not walked.  Every byte of the segment gets a source pointer -- itself
// (literal, or already copied) or the byte its deferred match names, in the block or in the block before --
// and pointer jumping (P[x] = P[P[x]]) resolves all chains at once, in log(depth) passes over the pointers;
// a last pass fetches each deferred byte from its root.  All of it is data-parallel over the whole segment:
// the rate no longer depends on how many streams there are.
//
// Pointer coordinates: index PTR_PRE + d is the byte d bytes behind `lo` in the output buffer (lo = start of
// the block before the segment's first block: its dictionary); indices below PTR_PRE are the last bytes of the
// caller's dictionary (dict0), which only block 0 of a call can reach.  The top bit marks a pointer that
// already names a root.
//
// Anything the reference would not have accepted (an offset beyond the dictionary, :1764; a dictionary match
// that ends inside the last literals, :1884-1889), and any block without a usable list (overflow, > 4 MiB), turns its STREAM down: its blocks
// are left untouched and walked by linked_replay.hpp / the exact decoder, which also yields the
// reference's error codes.  (DecodeArgs::ptrBad: one flag per stream.)
#pragma once

#include "decode_par.hpp"
#include "linked_replay.hpp"

namespace lz4dev {

#define PTR_FINAL 0x80000000u
#define PTR_PRE 65536u
// A pass follows up to PTR_JUMPS + 1 pointers from every unresolved byte and stores where it got to.  If every
// pointer spans at least s levels of its chain before a pass, it spans at least (PTR_JUMPS + 1) * s after it
// (each hop reads a pointer that is at least as good as before the pass), so k passes resolve every chain of
// depth < (PTR_JUMPS + 1)^k, whatever other threads have or have not written meanwhile.  A segment holds fewer
// than 2^31 pointers: PTR_MAX_PASSES passes always suffice, and the ones that find nothing to do cost 6 us each.
#ifndef PTR_JUMPS
#define PTR_JUMPS 11
#endif
constexpr int ptr_passes_for(int jumps)
{
    int k = 0;
    for (unsigned long long span = 1; span < (1ull << 31); span *= (unsigned long long)(jumps + 1)) k++;
    return k + 1;                      // + the pass that sees that nothing is left
}
#define PTR_MAX_PASSES (ptr_passes_for(PTR_JUMPS))
#ifndef PTR_PARTS
#define PTR_PARTS 16               // workgroups per block in the jump / fetch passes
#endif

struct PtrCtl {
    uint32_t changed[PTR_MAX_PASSES + 1];   // pass r left unresolved pointers behind
    uint32_t needOld;                       // linked_local.hpp gave something up: the pointer pass runs over the segment
};

// ---- local resolve + chase (round 3): the same second pass for blocks of up to 64 KiB, without a global pointer per byte ----
// The tolerant pass defers practically every match of a reference-written text stream, so the pointer pass above redoes
// the copying of the whole stream through 4-byte pointers in HBM.  Most of a chain's hops stay inside one block, though:
//   k_loc_resolve   one workgroup per dependent block.  16-bit source pointers of the block's bytes in LDS (128 KiB:
//                   one block per CU), pointer jumping there until every byte names its in-block root: either a byte the
//                   tolerant pass has already written (copied now), or a byte whose match starts in the block before --
//                   then all that is kept is WHERE in that block: its distance from the block's end (1..65535), one
//                   uint16 per output byte ("origins", 0 = nothing to do; they live in the pointer buffer).
//   k_loc_chase     every byte with an origin follows it: the byte it names is final (fetch it), or has an origin of its
//                   own in the block before that one -- block by block, reads only, until a final byte turns up.  Bytes of
//                   a match move together (consecutive origins): four at a time while they do.
// A chain that is followed through more than DecodeArgs::chaseMax blocks, or a dependent block above 64 KiB, raises
// PtrCtl::needOld and the pointer pass runs over the segment as before.
#define LOC_MAX 65536
#define LOC_THREADS 1024
#define LOC_HAS_ORIGINS 0x80000000u
#define LOC_NO_ORIGINS 0x40000000u
#ifndef LOC_HOPS
#define LOC_HOPS 4                 // pointers an item follows per round of the in-block jumping
#endif

struct __attribute__((aligned(16))) LocLds {
    uint16_t lp[LOC_MAX];          // source of byte i: a position of the block, or (ext bit set) a distance into the block before
    uint32_t ext[LOC_MAX / 32];    // byte i's match source lies in the block before (a root of the in-block jumping)
    int region[2];                 // positions the chunk of deferred matches in hand covers
};

} // namespace lz4dev
