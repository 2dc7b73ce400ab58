cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { shape="$1"; cfg="$2"; if [ "$cfg" = "default" ]; then out=$(python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1); else out=$(env $cfg python3 $R/tools/prof_cone.py $shape 2>/dev/null | tail -1); fi; echo "$shape | $cfg | $(echo $out | grep -o "'us': [0-9.]*\|'GBps': [0-9.]*" | tr '\n' ' ')"; }
for cfg in default "LEC_JOINT_GEOM=1,12,9" "LEC_JOINT_GEOM=1,16,9" "LEC_JOINT_GEOM=2,8,17" "LEC_JOINT_GEOM=2,8,9" "LEC_JOINT_GEOM=1,12,9 LEC_JOINT_STAGE=1" "LEC_JOINT_GEOM=4,4,33"; do run "4096 256 10 50000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=1,12,5" "LEC_JOINT_GEOM=1,12,3" "LEC_JOINT_GEOM=1,12,9"; do run "1024 256 10 50000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=1,12,3" "LEC_JOINT_GEOM=1,12,2"; do run "512 256 10 50000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=16,8,22" "LEC_JOINT_GEOM=16,8,11" "LEC_JOINT_GEOM=16,8,44"; do run "256 256 128 50000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=16,8,129" "LEC_JOINT_GEOM=16,8,33"; do run "4096 64 128 50000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=4,4,1" "LEC_JOINT_GEOM=1,12,1"; do run "256 5 10 2000" "$cfg"; done
for cfg in default "LEC_JOINT_GEOM=4,4,1" ; do run "4096 5 10 2000" "$cfg"; done
