# scratch/g1probe (1-wave kernel): checks + wall-clock timing, then one PMC pass for cycle counts per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
( cd scratch && timeout 300 ./g1probe ) 2>&1 | tail -30
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d /tmp/prof_g1 -o g1 -- $GRAFT_REPO_ROOT/scratch/g1probe time ) > $OUT/prof_g1.log 2>&1; echo "g1 pmc rc=$?"
python scripts/pmc_query.py $(find /tmp/prof_g1 -name "*.db" | head -1) > $OUT/pmc_g1.txt
python3 - <<'PY'
rows={}
for l in open('gpurun_out/pmc_g1.txt'):
    p=l.strip().split('|')
    if len(p)!=4: continue
    rows.setdefault(p[0],{})[p[1]]=float(p[2])
print("%-62s %9s %8s %8s %8s %8s"%("kernel","cyc/XCD","MFMAfrac","waitLDS","waitANY","wavecyc"))
for k,v in rows.items():
    if 'gemm' not in k: continue
    g=v.get('GRBM_GUI_ACTIVE',0)/8
    print("%-62s %9.0f %8.3f %8.2f %8.2f %8.2f"%(k[13:75], g, v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(g,1), v.get('SQ_WAIT_INST_LDS',0)/1e6, v.get('SQ_WAIT_INST_ANY',0)/1e6, v.get('SQ_WAVE_CYCLES',0)/1e6))
PY
