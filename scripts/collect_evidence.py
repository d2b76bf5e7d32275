#!/usr/bin/env python3
"""Copy one scripts/round_evidence.sh session from gpurun_out/ into profiles/ as rNN_* (default r03) and write
profiles/rNN_PROVENANCE.json (commit, files).    python scripts/collect_evidence.py [round]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
E = os.path.join(ROOT, "gpurun_out", "evidence")
P = os.path.join(ROOT, "profiles")
commit = open(os.path.join(E, "commit.txt")).read().strip()
for tag, workload, key in (("dec", "decompress", "decompress"), ("cmp", "compress", "compress"), ("text", "text", "decompress")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarize_profiles.py"), tag, workload, key, rnd])
names = {"bench_roundtrip.json": "bench_roundtrip.json", "bench_random256k.json": "bench_random256k.json",
         "bench_text_linked_compress.json": "bench_text_linked_compress.json", "bench_linked1.json": "bench_linked1.json",
         "bench_matrix_synthetic.jsonl": "bench_matrix_synthetic.jsonl", "encode_lzsynth_pmc_instmix.txt": "encode_lzsynth_pmc_instmix.txt",
         "encode_text_pmc_instmix.txt": "encode_text_pmc_instmix.txt", "decode_lzsynth_pmc_instmix.txt": "decode_lzsynth_pmc_instmix.txt",
         "decode_text_pmc_instmix.txt": "decode_text_pmc_instmix.txt", "small_batch_compress_latency.txt": "small_batch_compress_latency.txt",
         "small_call_decompress_breakdown.txt": "small_call_decompress_breakdown.txt", "linked_shard_split.txt": "linked_shard_split.txt",
         "linked_async_cost.txt": "linked_async_cost.txt", "linked_single_stream_kernel_stats.csv": "linked_single_stream_kernel_stats.csv", "bench_one_stream_rehearsal.jsonl": "bench_one_stream_rehearsal.jsonl", "realtext_ratio.txt": "realtext_ratio.txt", "size_vs_reference.txt": "size_vs_reference.txt",
         "host_api_rate.jsonl": "host_api_rate.jsonl", "valu_issue_rate.txt": "valu_issue_rate.txt",
         "bench_n2_selflaunch_rehearsal.json": "bench_n2_selflaunch_rehearsal.json", "linked_streams_rate.jsonl": "linked_streams_rate.jsonl", "kernel_resources.txt": "kernel_resources.txt",
         "bench_roundtrip_8GiB.json": "bench_roundtrip_8GiB.json", "linked_runin_decode.txt": "linked_runin_decode.txt",
         "cu_decode_small_calls.jsonl": "cu_decode_small_calls.jsonl", "cu_decode_kernel_stats.csv": "cu_decode_kernel_stats.csv",
         "cu_decode_lzsynth_pmc_instmix.txt": "cu_decode_lzsynth_pmc_instmix.txt", "cu_decode_text_pmc_instmix.txt": "cu_decode_text_pmc_instmix.txt",
         "decode_own_vs_reference_written.txt": "decode_own_vs_reference_written.txt", "multi_device_rehearsal.jsonl": "multi_device_rehearsal.jsonl",
         "cu_decode_crossover.txt": "cu_decode_crossover.txt", "runin_dictionary_share.txt": "runin_dictionary_share.txt", "cu_decode_lowratio.txt": "cu_decode_lowratio.txt", "big_linked_blocks.txt": "big_linked_blocks.txt"}
files = []
for src, dst in names.items():
    p = os.path.join(E, src)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(P, "%s_%s" % (rnd, dst)))
        files.append("%s_%s" % (rnd, dst))
for tag, workload in (("dec", "decompress"), ("cmp", "compress"), ("text", "text")):
    for suffix in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "pmc_fetch.csv", "pmc_write.csv"):
        f = "%s_bench_%s_%s" % (rnd, workload, suffix)
        if os.path.exists(os.path.join(P, f)):
            files.append(f)
json.dump({"round": rnd, "commit": commit, "session": "one gpurun call of scripts/round_evidence.sh (one box, one build)",
           "files": sorted(files)}, open(os.path.join(P, "%s_PROVENANCE.json" % rnd), "w"), indent=1)
print("collected", len(files), "files at commit", commit)
