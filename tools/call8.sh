#!/bin/bash
# round 6, GPU call 8: the whole GPU suite, smoke(), two-rank rehearsal over gloo on one GPU
set -o pipefail
mkdir -p gpurun_out/r06h
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/rehearse_dp2.py --single 2>&1 | tail -2
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rehearse_dp2.py 2>&1 | tail -4
