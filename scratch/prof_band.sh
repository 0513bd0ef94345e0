cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sh in ${SHAPES:-256x8192 512x8192 1024x8192 2048x4096}; do
  rm -rf /tmp/pb
  cat > /tmp/pb_run.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from bench import HotPathStep
B, Nc = (int(x) for x in "$sh".split("x"))
hp = HotPathStep(B, Nc // B, 768, 1.0, 1, 0, torch.device("cuda", 0))
for _ in range(60):
    hp.k_step()
torch.cuda.synchronize()
PY
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python /tmp/pb_run.py ) > gpurun_out/pb.log 2>&1
  echo "== $sh"
  python scripts/prof_summary.py band_$sh --trace "$(find /tmp/pb -name '*.db' | head -1)" --out gpurun_out/pb_sum | grep dprhot | cut -d, -f1-6 | cut -c1-170
done
