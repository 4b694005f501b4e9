#!/bin/bash
# GPU box: everything under profiles/<round>_* that depends on the kernels, in one go (about 6 GPU-minutes).
# Results land in gpurun_out/refresh/ under their profiles/ names; copy them over afterwards:
#   gpurun -- 'bash tools/refresh_profiles.sh r03' && cp gpurun_out/refresh/r03_* profiles/ && python tools/make_traffic_json.py r03 c2 c3 c4 c5
rnd=${1:-r03}
R=$GRAFT_REPO_ROOT/gpurun_out/refresh; rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
# kernel traces: the default regime (3 frames in flight) of every config, and c4 / c5 one frame at a time
for c in c4 c5 c2 c3; do
  bash tools/prof_bench.sh ${rnd}_$c $c > $R/${rnd}_${c}_kernel_trace_stats_default_regime.txt 2>&1 || exit 1
  cp gpurun_out/prof/${rnd}_$c/*/*_kernel_stats.csv $R/${rnd}_${c}_default_kernel_stats.csv
done
for c in c4 c5; do
  bash tools/prof_bench.sh ${rnd}_${c}_solo $c --frames-in-flight 1 > $R/${rnd}_${c}_kernel_trace_stats_one_frame_at_a_time.txt 2>&1 || exit 1
done
# HBM traffic per kernel (two --pmc passes each)
bash tools/prof_traffic.sh c4_torus200k_1080p c4 > $R/${rnd}_c4_pmc_fetch_write.txt 2>&1 || exit 1
bash tools/prof_traffic.sh c5_torus1m_4k_skybox c5 > $R/${rnd}_c5_pmc_fetch_write.txt 2>&1 || exit 1
bash tools/prof_traffic.sh c2_diablo_1080p c2 > $R/${rnd}_c2_pmc_fetch_write.txt 2>&1 || exit 1
bash tools/prof_traffic.sh c3_diablo_floor_1080p c3 > $R/${rnd}_c3_pmc_fetch_write.txt 2>&1 || exit 1
# SQ counters of the frame's kernels on c4 (two passes: the counters do not fit one)
{ bash tools/prof_pmc.sh sq1 k_ SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY
  bash tools/prof_pmc.sh sq2 k_ SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F64; } > $R/${rnd}_c4_pmc_sq_counters.txt 2>&1 || exit 1
# the vector-instruction mix of the frame's kernels (bench.py's roofline_valu) and what an instruction costs a SIMD
bash tools/prof_valu_mix.sh c4_torus200k_1080p > $R/${rnd}_c4_valu_mix.json 2> $R/valu_mix.err || exit 1
bash tools/prof_valu_mix.sh c5_torus1m_4k_skybox > $R/${rnd}_c5_valu_mix.json 2>> $R/valu_mix.err || exit 1
hipcc -O3 --offload-arch=gfx950 -o tools/micro/valu_f64_rate tools/micro/valu_f64_rate.hip && tools/micro/valu_f64_rate > $R/${rnd}_valu_rate.txt 2>&1 || exit 1
# the drop-in call: Scene.render() with a moving camera, overlay off and on, synchronous and pipelined
bash tools/prof_scene_render.sh > $R/${rnd}_c4_scene_render_kernel_trace.txt 2>&1 || exit 1
grep "c4_torus" gpurun_out/prof/scene_render/run.log > $R/${rnd}_c4_scene_render_host_breakdown.txt
# bench lines without the profiler (c4 with the CPU baseline: the line the driver records)
for c in c2 c3 c5; do timeout -k 10 600 python bench.py --config $c --no-cpu-baseline > $R/${rnd}_bench_$c.json 2> $R/bench_$c.err || exit 1; done
timeout -k 10 900 python bench.py > $R/${rnd}_bench_c4.json 2> $R/bench_c4.err || exit 1
# what one rank of a split frame costs this device (no collective), per partition; and the atomics micro-benchmark
python3 tools/time_band.py c4_torus200k_1080p 2 4 8 2>&1 | grep -v amdgpu.ids > $R/${rnd}_partition_per_rank_times.txt
python3 tools/time_band.py c5_torus1m_4k_skybox 2 4 8 2>&1 | grep -v amdgpu.ids >> $R/${rnd}_partition_per_rank_times.txt
PARTITIONS="stripes weighted" python3 tools/time_band.py c3_diablo_floor_1080p 2 4 8 2>&1 | grep -v amdgpu.ids >> $R/${rnd}_partition_per_rank_times.txt
hipcc -O3 --offload-arch=gfx950 -o tools/micro/atomic_same_addr tools/micro/atomic_same_addr.hip && tools/micro/atomic_same_addr > $R/${rnd}_atomic_same_address.txt 2>&1
# the raw traces stay on the box: gpurun copies back at most 64 MiB
rm -rf gpurun_out/prof gpurun_out/pmc
ls -la $R
