// san_stubs.cpp -- kernel launchers of kernels.hip, stubbed for the CPU-only sanitizer build of the host
// library (make asan / make tsan).  Never reached there: without a gfx950 device mi355lz4_create fails.
#include "../../streamly-lz4_amd/csrc/kernels.h"

#include <cstdlib>

#define UNREACHABLE_LAUNCH abort()
void launch_decode_seq(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_decode_par(const DecodeArgs &, unsigned long long *, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_decode_cu(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_dict_share(const DecodeArgs &, int, int, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_cu_linked(const DecodeArgs &, bool, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_cu_publish(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
#ifdef MI355LZ4_EXPERIMENTS
void launch_decode_tok(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
#endif
void launch_linked_tolerant(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_linked_resolve(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_linked_resolve_a(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_linked_resolve_b(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_linked_fetch_block(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
size_t ptr_ctl_last_open_offset() { return 0; }
void launch_longest_stream(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_linked_runs(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_link_stat(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_runin_decode(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_runin_fix(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_runin_publish(const DecodeArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
size_t tol_region_bytes() { return 65536; }
void launch_encode(const EncodeArgs &, bool, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_encode_seg(const EncodeSegArgs &, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_compact(const uint8_t *, size_t, const int32_t *, int, uint8_t *, size_t, uint64_t *, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_interleave(const uint8_t *, const uint64_t *, int, int, int, uint8_t *, const uint64_t *, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_index(const uint8_t *, uint64_t, const uint64_t *, int, int, int, int32_t *, uint64_t *, hipStream_t) { UNREACHABLE_LAUNCH; }
void launch_generate(int, uint8_t *, int, int, uint64_t, uint64_t, uint32_t, uint32_t, hipStream_t) { UNREACHABLE_LAUNCH; }
size_t ptr_ctl_bytes() { return 64; }
