cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/bench_rankstep.py --shapes 32:8:768:1,32:8:768:2,32:8:768:4,32:8:768:8,64:2:1024:8,128:8:768:8 --reps 100 2>&1 | grep "^{" | cut -c1-170
echo "no skinny:"; DPRHOT_NO_SKINNY=1 python scripts/bench_rankstep.py --shapes 32:8:768:8 --reps 100 2>&1 | grep "^{" | cut -c1-170
rm -rf /tmp/prof_rank
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rank -o rank -- python $GRAFT_REPO_ROOT/scripts/bench_rankstep.py --shapes 32:8:768:8 --eager --reps 50 ) > gpurun_out/prof_rank32.log 2>&1
python scripts/prof_summary.py r02_cfg2w8rank --trace $(find /tmp/prof_rank -name "*.db" | head -1) --out gpurun_out/prof_summary | cut -c1-130 | head -7
