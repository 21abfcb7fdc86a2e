#!/bin/bash
# dev: everything round 5 commits under profiles/ (run on the GPU box):  bash tools_dev/session_r05.sh
cd $GRAFT_REPO_ROOT
bash tools_dev/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
for c in C2 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r05_bench_$c.json 2> gpurun_out/err_$c.txt; done
python bench.py --config C5 --pipeline --steps 8 --warmup 2 > gpurun_out/r05_bench_C5_pipeline.json 2> gpurun_out/err_p5.txt
python bench.py --config C5 --pipeline --steps 8 --warmup 2 --motion wild > gpurun_out/r05_bench_C5_pipeline_wild.json 2> gpurun_out/err_p5w.txt
python bench.py --config C4 --pipeline --steps 8 --warmup 2 > gpurun_out/r05_bench_C4_pipeline.json 2> gpurun_out/err_p4.txt
python bench.py --config C4 --pipeline --steps 8 --warmup 2 --motion wild > gpurun_out/r05_bench_C4_pipeline_wild.json 2> gpurun_out/err_p4w.txt
python bench.py --config LVD --steps 200 --warmup 20 > gpurun_out/r05_bench_LVD.json 2> gpurun_out/err_lvd.txt
python bench.py --config LVD --steps 200 --warmup 20 --graph > gpurun_out/r05_bench_LVD_graph.json 2> gpurun_out/err_lvdg.txt
python tools_dev/glue_ops.py LVD > gpurun_out/r05_lvd_step_framework_ops.txt 2> gpurun_out/err_glue.txt
python tools_dev/lvd_host_profile.py > gpurun_out/r05_lvd_step_host_profile.txt 2>&1
# one job split over ranks (bench.py --scaling strong): every rank's share timed on this one GPU, eager and from one HIP graph
python tools_dev/strong_projection.py C5 > gpurun_out/r05_strong_projection_C5.json 2> gpurun_out/err_sp5.txt
python tools_dev/strong_projection.py C4 > gpurun_out/r05_strong_projection_C4.json 2> gpurun_out/err_sp4.txt
# the two north-star command lines with two ranks sharing this GPU (gloo): that the lines run, not a scaling number
python bench.py --config C5 --pipeline --gpus 2 --scaling strong --dist-backend gloo --steps 4 --warmup 1 > gpurun_out/r05_bench_C5_pipeline_strong_2ranks_one_gpu.json 2> gpurun_out/err_s5.txt
python bench.py --config C4 --pipeline --gpus 2 --scaling strong --graph --dist-backend gloo --steps 4 --warmup 1 > gpurun_out/r05_bench_C4_pipeline_strong_graph_2ranks_one_gpu.json 2> gpurun_out/err_s4.txt
bash tools_dev/pmc_any.sh r05_C5 bench.py --config C5 --pipeline --steps 2 --warmup 1 > gpurun_out/r05_pipeline_C5_counters.txt 2>&1
bash tools_dev/pmc_any.sh r05_LVD bench.py --config LVD --steps 10 --warmup 2 > gpurun_out/r05_lvd_step_counters.txt 2>&1
cp gpurun_out/pmc_r05_C5/stats/*/*kernel_stats.csv gpurun_out/r05_pipeline_C5_kernel_stats.csv
cp gpurun_out/pmc_r05_LVD/stats/*/*kernel_stats.csv gpurun_out/r05_lvd_step_kernel_stats.csv
[ -x tools_dev/r3_stream ] || hipcc --offload-arch=gfx950 -O3 -o tools_dev/r3_stream tools_dev/r3_stream.hip
./tools_dev/r3_stream > gpurun_out/r05_micro_stream.txt 2>&1
echo done
