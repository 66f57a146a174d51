"""MI355X engine behind the reference's method classes.

Independent reference batches of one call run on up to three HIP streams.  HIP maps the streams of a process onto
four hardware queues by default, and RCCL's internal streams occupy some of them once a torch.distributed process
group exists; streams that share a queue serialise (K=100 bench under torch.distributed.run: 1 641 tasks/s against
2 119).  Eight queues restore the rate.  The variable is read when the HIP runtime initialises, i.e. at the first
GPU call of the process, so the default is set here at import time; an explicit setting of the user wins."""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
