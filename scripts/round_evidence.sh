#!/bin/bash
# Everything profiles/rNN_* is made of, in one session on one box at one commit:
#   PROFILE_COMMIT=<short hash> bash scripts/round_evidence.sh
# Results land under gpurun_out/evidence/ (and gpurun_out/prof_<tag>/ for the three profiled workloads);
# scripts/collect_evidence.py then copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
E=$R/gpurun_out/evidence
rm -rf "$E"; mkdir -p "$E"
echo "${PROFILE_COMMIT:-}" > "$E/commit.txt"
cd "$R"
bash scripts/profile_bench.sh dec > "$E/profile_dec.log" 2>&1
bash scripts/profile_bench.sh cmp --workload compress > "$E/profile_cmp.log" 2>&1
bash scripts/profile_bench.sh text --workload text > "$E/profile_text.log" 2>&1
for k in lzsynth text; do bash scripts/pmc_encode.sh $k > "$E/encode_${k}_pmc_instmix.txt" 2>&1; done
for k in lzsynth text; do bash scripts/pmc_decode.sh $k > "$E/decode_${k}_pmc_instmix.txt" 2>&1; done
python3 bench.py --workload roundtrip --steps 10 --warmup 2 --no-cpu-baseline --no-host-api > "$E/bench_roundtrip.json" 2> "$E/bench_roundtrip.err"
# BASELINE config 4's per-GPU share (64 GiB / 8): 131 072 blocks, one call each way
python3 bench.py --workload roundtrip --blocks 131072 --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > "$E/bench_roundtrip_8GiB.json" 2> "$E/bench_roundtrip_8GiB.err"
python3 bench.py --workload random256k --steps 10 --warmup 2 --no-cpu-baseline --no-host-api > "$E/bench_random256k.json" 2> "$E/bench_random256k.err"
python3 bench.py --workload text --steps 20 --warmup 2 --linked-compress --no-cpu-baseline > "$E/bench_text_linked_compress.json" 2> "$E/bench_text_linked_compress.err"
python3 bench.py --steps 20 --warmup 2 --linked --no-cpu-baseline --no-host-api --no-extra > "$E/bench_linked1.json" 2> "$E/bench_linked1.err"
python3 bench_matrix.py --synthetic --cpu > "$E/bench_matrix_synthetic.jsonl" 2> "$E/bench_matrix.err"
python3 scripts/small_enc_latency.py 2>/dev/null | grep -v amdgpu > "$E/small_batch_compress_latency.txt"
python3 scripts/small_dec_breakdown.py 2>/dev/null | grep accel > "$E/small_call_decompress_breakdown.txt"
python3 scripts/linked_shard_split.py 4096 2>/dev/null | tail -1 > "$E/linked_shard_split.txt"
python3 scripts/linked_shard_split.py 1024 2>/dev/null | tail -1 >> "$E/linked_shard_split.txt"
python3 bench.py --one-stream --workload text --steps 5 --warmup 1 --blocks 4096 > "$E/bench_one_stream_rehearsal.jsonl" 2>/dev/null
for N in 2 3; do
  BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N \
      bench.py --gpus $N --steps 5 --warmup 1 --workload text --one-stream --blocks 2048 2>/dev/null | grep '^{' >> "$E/bench_one_stream_rehearsal.jsonl"
done
( cd /tmp && export TMPDIR=/tmp && rm -rf "$E/linked_prof" && rocprofv3 --kernel-trace --stats --output-format csv -d "$E/linked_prof" -- python3 "$R/scripts/prof_linked.py" 4096 5 > "$E/linked_prof.log" 2>&1; cp "$E"/linked_prof/*/*kernel_stats.csv "$E/linked_single_stream_kernel_stats.csv" 2>/dev/null )
# one reference-written linked text stream of 1, 2 and 4 GiB (and 1 GiB in blocks of 256 KiB and 1 MiB): pointer pass against run-in decode
{ for nb in 16384 32768 65536; do RUNIN_CFGS=9:0,13:0 python3 scripts/linked_runin_time.py text $nb 2>/dev/null | grep -v amdgpu; done
  python3 scripts/linked_runin_time.py text 4096 262144 2>/dev/null | grep -v amdgpu
  python3 scripts/linked_runin_time.py text 1024 1048576 2>/dev/null | grep -v amdgpu; } > "$E/linked_runin_decode.txt"
python3 scripts/linked_async_cost.py 2>/dev/null | grep blocks > "$E/linked_async_cost.txt"
python3 scripts/realtext_ratio.py 2>/dev/null | grep input > "$E/realtext_ratio.txt"
python3 scripts/size_vs_ref.py 2>/dev/null | grep segs > "$E/size_vs_reference.txt"
python3 scripts/host_api_rate.py > "$E/host_api_rate.jsonl" 2>/dev/null
rm -f gpurun_out/linked_rate.json; python3 -m pytest tests/test_linked_rate_gpu.py -q -m gpu > "$E/linked_rate_test.log" 2>&1
rm -f gpurun_out/linked_rate.json.prev; cp gpurun_out/linked_rate.json "$E/linked_streams_rate.jsonl" 2>/dev/null
scripts/kernel_resources.sh > "$E/kernel_resources.txt" 2>&1
# round 6: calls that do not fill the GPU (one workgroup per block, csrc/decode_cu.hpp): call times and the first segment's phases,
# the kernel's rocprofv3 stats and counters, the engine's against the reference's stream in calls of one size, the multi-device handle
{ python3 scripts/cu_decode_check.py time 2>/dev/null | grep '^{'
  python3 scripts/cu_decode_check.py time blocks=16 bl=262144 2>/dev/null | grep '^{'
  python3 scripts/cu_decode_check.py time blocks=256 bl=1048576 2>/dev/null | grep '^{'
  python3 scripts/cu_decode_check.py time blocks=3 bl=4194304 2>/dev/null | grep '^{'; } > "$E/cu_decode_small_calls.jsonl"
( cd /tmp && export TMPDIR=/tmp && rm -rf "$E/cu_prof" && rocprofv3 --kernel-trace --stats --output-format csv -d "$E/cu_prof" -- python3 "$R/scripts/prof_cu.py" lzsynth 160 30 > "$E/cu_prof.log" 2>&1; cp "$E"/cu_prof/*/*kernel_stats.csv "$E/cu_decode_kernel_stats.csv" 2>/dev/null )
for k in lzsynth text; do bash scripts/pmc_cu.sh $k > "$E/cu_decode_${k}_pmc_instmix.txt" 2>&1; done
python3 scripts/par_stats_ref.py lzsynth 16384 2>/dev/null | grep -v amdgpu > "$E/decode_own_vs_reference_written.txt"
python3 scripts/multi_device_rate.py 256 2>/dev/null | grep '^{' > "$E/multi_device_rehearsal.jsonl"
# which calls take the workgroup form (api.cpp, cu_auto): both forms at the call shapes either side of the rule
{ for n in 256 512 768; do echo "== 64 KiB blocks, $n a call"; python3 scripts/cu_decode_check.py time brief blocks=$n 2>/dev/null | grep -v amdgpu | cut -c1-72; done
  for n in 256 512; do echo "== 16 KiB blocks, $n a call"; python3 scripts/cu_decode_check.py time brief blocks=$n bl=16384 2>/dev/null | grep -v amdgpu | cut -c1-72; done
  echo "== 4 KiB blocks, 160 a call"; python3 scripts/cu_decode_check.py time brief blocks=160 bl=4096 2>/dev/null | grep -v amdgpu | cut -c1-72; } > "$E/cu_decode_crossover.txt"
# the dictionary share a linked call samples before it picks its run-in (api.cpp, k_dict_share), for the streams it has to tell apart
python3 scripts/runin_share.py 2>/dev/null | grep share > "$E/runin_dictionary_share.txt"
# linked streams of big blocks: the workgroup form against guessed dictionaries (api.cpp path 6) and the pointer pass
python3 scripts/big_linked_rate.py 2>/dev/null | grep linked > "$E/big_linked_blocks.txt"
# the two decoder forms and the default choice on streams that hardly compress, and on very compressible ones
python3 scripts/cu_decode_lowratio.py 2>/dev/null | grep -v amdgpu > "$E/cu_decode_lowratio.txt"
# issue rate of integer vector instructions (section 0 of DESIGN.md prices the decoder with these)
[ -x scripts/micro/valu_rate.bin ] && scripts/micro/valu_rate.bin > "$E/valu_issue_rate.txt" 2>&1
# `python3 bench.py --gpus 2` on its own (the script starts its ranks as a child process), rehearsed on this one GPU over gloo
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --blocks 4096 \
    > "$E/bench_n2_selflaunch_rehearsal.json" 2> "$E/bench_n2_selflaunch_rehearsal.err"; echo "rc=$?" >> "$E/bench_n2_selflaunch_rehearsal.err"
grep -l "Memory access fault" "$E"/* && exit 9
ls -la "$E" | tail -40
