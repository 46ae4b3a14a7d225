cd $GRAFT_REPO_ROOT
TE_DIST_WORLD1_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python -c "
from traversability_estimation_amd import dist as d
import torch
d.init_process_group('nccl'); print('max', d.max_over_ranks(1.5)); d.barrier()
import torch.distributed as td; td.destroy_process_group()
" > /tmp/o.txt 2> /tmp/e.txt
echo "--- stdout"; cat /tmp/o.txt; echo "--- stderr (head)"; head -5 /tmp/e.txt
echo "--- with NCCL_DEBUG unset, RCCL_* env:"; env | grep -i "nccl\|rccl" | head
