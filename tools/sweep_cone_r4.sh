# round 4: fused cone-loss kernel at the two K = 256 stress shapes: lane-per-pair (T=1) direct vs LDS-staged vs 4 lanes per pair, a process per setting
# (LEC_JOINT_STAGE is read once per process), and the counters rocprofv3 offers for the L2 atomic / wait picture.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "256 256 10 50000" "4096 256 10 50000"; do
  for cfg in "default" "LEC_JOINT_STAGE=1" "LEC_JOINT_GEOM=4,4,0" "LEC_JOINT_GEOM=4,4,2" "LEC_JOINT_GEOM=4,4,6" "LEC_JOINT_GEOM=2,8,0" "LEC_JOINT_GEOM=1,12,3" "LEC_JOINT_GEOM=1,12,12"; do
    if [ "$cfg" = "default" ]; then out=$(python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1); else out=$(env $cfg python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1); fi
    echo "$shape | $cfg | $out"
  done
done
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\|TCC_EA0_[A-Z_]*\|SQ_WAIT_[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVES\|SQ_INSTS_VALU\|SQ_INSTS_SALU\|SQ_INSTS_LDS\|SQ_INSTS_VMEM_WR\|SQ_INSTS_VMEM_RD\|SQ_ACTIVE_INST_[A-Z_]*\|TCP_TCC_ATOMIC[A-Z_]*\|TCP_[A-Z_]*ATOMIC[A-Z_]*\|TCC_REQ_sum\|TCC_HIT_sum\|TCC_MISS_sum\|GRBM_GUI_ACTIVE\)\b" | sort -u | tr '\n' ' '
