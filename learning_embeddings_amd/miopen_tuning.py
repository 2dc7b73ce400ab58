"""Ships the MIOpen find-db / perf-db records tuned on an MI355X for the conv shapes of the bench workloads, so that a
fresh box does not spend minutes benchmarking every solver (including MIOpen's naive reference convolutions, ~0.5 s per
launch) before the first step.  The records are data produced by MIOpen itself on the target GPU (`bench.py
--cudnn-benchmark` with MIOPEN_USER_DB_PATH pointing at an empty directory); shapes that are not in the db are tuned
at first use as usual.  Must be called BEFORE the first convolution runs (MIOpen reads the db path once)."""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))


def setup():
    if os.environ.get('MIOPEN_USER_DB_PATH'):
        return os.environ['MIOPEN_USER_DB_PATH']          # the user manages their own db
    src = os.path.join(_HERE, 'miopen_db')
    dst = os.path.join(tempfile.gettempdir(), 'lec_miopen_db_%d' % os.getuid())
    os.makedirs(dst, exist_ok=True)
    if os.path.isdir(src):
        for f in os.listdir(src):
            target = os.path.join(dst, f)
            if not os.path.exists(target):                # keep records a previous run on this box has added
                tmp = '%s.%d.tmp' % (target, os.getpid())  # atomic: several ranks start at once
                shutil.copy(os.path.join(src, f), tmp)
                os.replace(tmp, target)
    os.environ['MIOPEN_USER_DB_PATH'] = dst
    return dst
