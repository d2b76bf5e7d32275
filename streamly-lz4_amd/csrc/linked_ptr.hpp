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
    uint32_t lastOpen;                      // the fetch of one block alone (DecodeArgs::onlyBlk) left bytes of it unresolved
};

} // namespace lz4dev
