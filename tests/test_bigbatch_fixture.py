"""CPU: the lean big-batch fixture (tests/golden/bigbatch_*.npz, made by running the reference on a 170-task batch) is
reproducible here - its inputs come back from the integer generator bit for bit - and the C++ oracle takes the
reference's first outer iteration on it: the early stop at MM iteration 151 of a 17 000-row batch and the same argmax.
(The whole 20 x 1000 schedule is the GPU test's job, tests/test_gpu_round3.py.)"""
import hashlib
import os

import numpy as np

from conftest import GOLDEN
from helpers import intsynth
from oracle import c_oracle


def test_bigbatch_fixture_inputs_and_first_iteration():
    g = np.load(os.path.join(GOLDEN, "bigbatch_zs_soft_K100_N170.npz"))
    K, N = int(g["K"]), int(g["N"])
    assert N * K > 16384 and str(g["inputs"]) == "intsynth"
    x_q, y_q = intsynth.make_tasks(int(g["seed"]), N, K, 75, boost=int(g["boost"]))
    assert hashlib.sha1(np.ascontiguousarray(x_q).tobytes()).hexdigest() == str(g["x_q_sha1"])
    assert np.array_equal(y_q, g["y_q"].reshape(N, 75))
    assert g["mm_iters"][0] == 151 and (g["mm_iters"][1:] == 1000).all()       # one early stop, 19 x 19 decisions not to
    ref = c_oracle.run(x_q, iters=1, iter_mm=1000, lambd=int(K / 5) * 75)
    assert ref["mm_iters"][0] == g["mm_iters"][0]
    assert np.array_equal(ref["argmax"][0], g["argmax"][0])
    assert ref["criterions"][0] == g["criterions"][0]


def test_round4_bigbatch_fixtures_inputs_are_reproducible():
    """the two lean fixtures of round 4 (HARD at K = 397, FEW-SHOT at K = 100; both over 16 384 rows): the integer
    generator returns their inputs bit for bit on this host, the recorded MM counts have the reference's pattern, and the
    hard one contains an early stop decided over the whole 16 674-row batch.  (The schedules themselves are the GPU
    tests' job, tests/test_gpu_round4.py.)"""
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()      # noqa: E731
    g = np.load(os.path.join(GOLDEN, "bigbatch_zs_hard_K397_N42.npz"))
    K, N = int(g["K"]), int(g["N"])
    assert N * K > 16384 and str(g["kind"]) == "zs_hard" and str(g["inputs"]) == "intsynth"
    x_q, y_q = intsynth.make_tasks(int(g["seed"]), N, K, 75, boost=int(g["boost"]))
    assert sha(x_q) == str(g["x_q_sha1"]) and np.array_equal(y_q, g["y_q"].reshape(N, 75))
    assert g["mm_iters"].tolist() == [101] + [1000] * 9
    g = np.load(os.path.join(GOLDEN, "bigbatch_fs_soft_K100_N170_s1.npz"))
    K, N = int(g["K"]), int(g["N"])
    assert N * K > 16384 and str(g["kind"]) == "fs_soft" and int(g["shots"]) == 1
    x_q, y_q, x_s, y_s = intsynth.make_tasks(int(g["seed"]), N, K, 75, shots=1, boost=int(g["boost"]))
    assert sha(x_q) == str(g["x_q_sha1"]) and sha(x_s) == str(g["x_s_sha1"])
    assert np.array_equal(y_q, g["y_q"].reshape(N, 75)) and np.array_equal(y_s, g["y_s"].reshape(N, K))
    assert (g["mm_iters"] == 1000).all()            # 20 x 19 recorded decisions not to stop, each over 17 000 rows


def test_round5_k1000_bigbatch_fixture_first_iteration():
    """`bigbatch_zs_soft_K1000_N17` (round 5; 17 tasks x K = 1000 = 17 000 rows, made by running the reference for 6 700 s): the
    integer generator returns its inputs bit for bit, the recorded MM counts have the reference's pattern with an early stop
    decided over the whole batch at l = 100, and the C++ oracle takes the reference's first outer iteration on it - the stop at
    MM iteration 101, the same criterion, the same argmax.  (The 20 x 1000 schedule on the headline's kernels is the GPU
    test's job, tests/test_gpu_round5.py.)"""
    g = np.load(os.path.join(GOLDEN, "bigbatch_zs_soft_K1000_N17.npz"))
    K, N = int(g["K"]), int(g["N"])
    assert (K, N) == (1000, 17) and N * K > 16384 and str(g["inputs"]) == "intsynth" and int(g["iters"]) == 20
    x_q, y_q = intsynth.make_tasks(int(g["seed"]), N, K, 75, boost=int(g["boost"]))
    assert hashlib.sha1(np.ascontiguousarray(x_q).tobytes()).hexdigest() == str(g["x_q_sha1"])
    assert np.array_equal(y_q, g["y_q"].reshape(N, 75))
    assert g["mm_iters"].tolist() == [101] + [1000] * 19
    ref = c_oracle.run(x_q, iters=1, iter_mm=1000, lambd=int(K / 5) * 75)
    assert ref["mm_iters"][0] == 101
    assert np.array_equal(ref["argmax"][0], g["argmax"][0])
    assert ref["criterions"][0] == g["criterions"][0]
