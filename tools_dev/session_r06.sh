#!/bin/bash
# dev: everything round 6 commits under profiles/ (run on the GPU box):  bash tools_dev/session_r06.sh
# in two parts (a call of gpurun is limited to 20 minutes):  bash tools_dev/session_r06.sh a   /   ... b
cd $GRAFT_REPO_ROOT
if [ "${1:-a}" = "a" ]; then
bash tools_dev/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
# two more runs of the driver's command on the same box: the robustness the verdict asked for (within 1 %)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_driver_cmd_run1.json 2> gpurun_out/err_d1.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_driver_cmd_run2.json 2> gpurun_out/err_d2.txt
for c in C2 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r06_bench_$c.json 2> gpurun_out/err_$c.txt; done
python bench.py --config WIF --steps 50 --warmup 5 > gpurun_out/r06_bench_WIF.json 2> gpurun_out/err_wif.txt
python bench.py --config C5 --pipeline --steps 10 --warmup 3 > gpurun_out/r06_bench_C5_pipeline.json 2> gpurun_out/err_p5.txt
python bench.py --config C5 --pipeline --steps 10 --warmup 3 --motion wild > gpurun_out/r06_bench_C5_pipeline_wild.json 2> gpurun_out/err_p5w.txt
python bench.py --config C4 --pipeline --steps 10 --warmup 3 > gpurun_out/r06_bench_C4_pipeline.json 2> gpurun_out/err_p4.txt
python bench.py --config C4 --pipeline --steps 10 --warmup 3 --motion wild > gpurun_out/r06_bench_C4_pipeline_wild.json 2> gpurun_out/err_p4w.txt
python bench.py --config LVD --steps 200 --warmup 20 > gpurun_out/r06_bench_LVD.json 2> gpurun_out/err_lvd.txt
python bench.py --config LVD --steps 200 --warmup 20 --graph > gpurun_out/r06_bench_LVD_graph.json 2> gpurun_out/err_lvdg.txt
python tools_dev/lvd_host_profile.py > gpurun_out/r06_lvd_step_host_profile.txt 2>&1
echo done a
else
# one job split over ranks: every rank's share timed on this one GPU, eager and from one HIP graph
python tools_dev/strong_projection.py C5 > gpurun_out/r06_strong_projection_C5.json 2> gpurun_out/err_sp5.txt
python tools_dev/strong_projection.py C4 > gpurun_out/r06_strong_projection_C4.json 2> gpurun_out/err_sp4.txt
# the driver's multi-GPU command with two ranks sharing this GPU (gloo): the line with its north_star block
python bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 > gpurun_out/r06_bench_2ranks_one_gpu_north_star.json 2> gpurun_out/err_ns.txt
# kernel stats + HBM-side traffic + issue counters of the WIF training step and the C5 pipeline
bash tools_dev/pmc_any.sh r06_WIF bench.py --config WIF --steps 3 --warmup 1 > gpurun_out/r06_wif_step_counters.txt 2>&1
cp gpurun_out/pmc_r06_WIF/stats/*/*kernel_stats.csv gpurun_out/r06_wif_step_kernel_stats.csv; rm -rf gpurun_out/pmc_r06_WIF
bash tools_dev/pmc_any.sh r06_C5 bench.py --config C5 --pipeline --steps 2 --warmup 1 > gpurun_out/r06_pipeline_C5_counters.txt 2>&1
cp gpurun_out/pmc_r06_C5/stats/*/*kernel_stats.csv gpurun_out/r06_pipeline_C5_kernel_stats.csv; rm -rf gpurun_out/pmc_r06_C5
bash tools_dev/pmc_any.sh r06_LVD bench.py --config LVD --steps 10 --warmup 2 > gpurun_out/r06_lvd_step_counters.txt 2>&1
cp gpurun_out/pmc_r06_LVD/stats/*/*kernel_stats.csv gpurun_out/r06_lvd_step_kernel_stats.csv; rm -rf gpurun_out/pmc_r06_LVD
[ -x tools_dev/r3_stream ] || hipcc --offload-arch=gfx950 -O3 -o tools_dev/r3_stream tools_dev/r3_stream.hip
./tools_dev/r3_stream > gpurun_out/r06_micro_stream.txt 2>&1
echo done b
fi
