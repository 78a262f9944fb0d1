export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not emulated and not resnet and not leaf_symmetry and not all_sizes" > gpurun_out/r03b_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r03b_suite.log
for sm in 32 64 128; do AZ_SPLIT_MAX=$sm python3 tools/latency_prof.py search; AZ_SPLIT_MAX=$sm python3 tools/latency_prof.py arena; done 2>&1 | grep -v amdgpu.ids
AZ_TILE_SPLIT=0 AZ_SPLIT_MAX=64 python3 tools/latency_prof.py arena 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03b_lat_search -- python3 tools/latency_prof.py search > gpurun_out/prof/r03b_lat_search.log 2>&1
python3 tools/kstats.py gpurun_out/prof/r03b_lat_search_kernel_stats.csv 8
AZ_SPLIT_MAX=64 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03b_lat_arena -- python3 tools/latency_prof.py arena > gpurun_out/prof/r03b_lat_arena.log 2>&1
python3 tools/kstats.py gpurun_out/prof/r03b_lat_arena_kernel_stats.csv 10
