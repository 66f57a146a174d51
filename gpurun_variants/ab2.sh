cd $GRAFT_REPO_ROOT
for v in opt2; do echo "== selftest $v"; TCLIP_LIB=$PWD/gpurun_variants/$v.so timeout 600 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu 2>&1 | tail -3; done
timeout 1500 python scripts/gpu_ab_libs.py gpurun_variants/base.so gpurun_variants/tabns.so gpurun_variants/opt2.so -- 1000 3 125 20 0 0 100 10 100 20 0 0 397 4 100 10 1 0 1000 2 25 20 0 1
