cd $GRAFT_REPO_ROOT
run() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],1))"; }
python bench.py --workload k100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | run plain
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --workload k100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | run torchrun
GPU_MAX_HW_QUEUES=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --workload k100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | run torchrun_q8
GPU_MAX_HW_QUEUES=8 python bench.py --workload k100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | run plain_q8
TCLIP_STREAM_GROUPS=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29536 bench.py --gpus 1 --workload k100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | run torchrun_g2
