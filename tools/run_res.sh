export TMPDIR=/tmp
AZ_ENGINE_LIB=$PWD/alphazero-piskvorky_amd/libaz_engine_stamps.so python tools/stamps.py 15 2>&1 | grep -v amdgpu > gpurun_out/r03_stamps_f32.txt; cat gpurun_out/r03_stamps_f32.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "resnet or leaf_symmetry or all_sizes" > gpurun_out/r03e_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r03e_suite.log
python3 tools/latency.py gpurun_out/r03_latency.json 2>&1 | grep -v amdgpu
