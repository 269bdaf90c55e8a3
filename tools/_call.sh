cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3x8; mkdir -p $O
RICADI_TIMING=1 RICADI_DEBUG_FREE=1 python bench.py --no-large-roofline --no-cpu-baseline --no-extras --steps 3 --warmup 1 2> $O/free.err > /dev/null
python - <<'PY'
import re,collections
lines=open("gpurun_out/r3x8/free.err").read().splitlines()
# indices of 'Newton step 1: total' lines
idx=[i for i,l in enumerate(lines) if "Newton step" in l and "total" in l]
print(len(idx), "newton lines")
# frees between the last two Newton lines (one full timed step)
a,b=idx[-2],idx[-1]
c=collections.Counter(l.split()[2] for l in lines[a:b] if "ricadi free" in l)
print(sum(c.values()), "frees in one step:", c.most_common(12))
PY
