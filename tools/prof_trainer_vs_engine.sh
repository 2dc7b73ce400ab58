# kernel traces of the eager fp32 step: synthetic-input engine (504 rows) against the trainer API (about 504 rows), same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt_eng $R/gpurun_out/kt_tr
rocprofv3 --kernel-trace -d $R/gpurun_out/kt_eng -o t -- python3 $R/tools/engine_eager_probe.py --batch 252 > $R/gpurun_out/kt_eng.out 2> $R/gpurun_out/kt_eng.err
python3 $R/tools/summarize_rocpd.py $(ls $R/gpurun_out/kt_eng/*/*.db $R/gpurun_out/kt_eng/*.db 2>/dev/null | head -1) --steps 6 --skip-last 1 > $R/gpurun_out/kt_eng_summary.md 2>> $R/gpurun_out/kt_eng.err
rocprofv3 --kernel-trace -d $R/gpurun_out/kt_tr -o t -- python3 $R/tools/trainer_alloc_probe.py > $R/gpurun_out/kt_tr.out 2> $R/gpurun_out/kt_tr.err
python3 $R/tools/summarize_rocpd.py $(ls $R/gpurun_out/kt_tr/*/*.db $R/gpurun_out/kt_tr/*.db 2>/dev/null | head -1) --steps 6 --skip-last 1 > $R/gpurun_out/kt_tr_summary.md 2>> $R/gpurun_out/kt_tr.err
rm -rf $R/gpurun_out/kt_eng $R/gpurun_out/kt_tr
grep "^pure" $R/gpurun_out/kt_eng.out; grep "^run" $R/gpurun_out/kt_tr.out
head -30 $R/gpurun_out/kt_eng_summary.md; head -30 $R/gpurun_out/kt_tr_summary.md
