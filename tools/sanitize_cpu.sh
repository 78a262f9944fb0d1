#!/bin/bash
# CPU-side sanitizer pass (GPU ASan is not available on the pool): the oracle under ASan+UBSan through its golden tests,
# and the engine's host RNG (az_rng.cpp) under ASan+UBSan against numpy's own RandomState streams.
set -e
cd "$(dirname "$0")/.."
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
make -C oracle liboracle_asan.so
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  AZ_ORACLE_LIB=$PWD/oracle/liboracle_asan.so python -m pytest tests/test_oracle_golden.py tests/test_resnet_cpu.py tests/test_leaf_symmetry_cpu.py -x -q
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -o /tmp/libaz_rng_asan.so \
  alphazero-piskvorky_amd/csrc/az_rng.cpp -lpthread
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python3 - <<'PY'
import ctypes as C, numpy as np
L = C.CDLL("/tmp/libaz_rng_asan.so")
L.az_rng_selfplay_tape.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
for n, seed, alpha in ((5, 7, 0.3), (9, 123456789, 0.3), (15, 2**31 + 5, 1.7), (3, 0, 1.0)):
    nn = n * n
    noise = np.zeros(nn * (nn + 1) // 2); u = np.zeros(nn)
    assert L.az_rng_selfplay_tape(seed, n, alpha, 0, noise.ctypes.data, u.ctypes.data) == 0
    rs = np.random.RandomState(seed & 0xffffffff)
    ref_n, ref_u = [], []
    for m in range(nn):
        ref_n.append(rs.dirichlet([alpha] * (nn - m))); ref_u.append(rs.random_sample())
    assert np.array_equal(noise, np.concatenate(ref_n)) and np.array_equal(u, np.array(ref_u)), (n, seed)
print("az_rng under ASan/UBSan: numpy-identical, clean")
PY
