# On the GPU box (gpurun -- bash tools/prof_ball_query.sh): rocprofv3 kernel statistics of one bench / microbenchmark command, top kernels printed;
# the csv lands in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bqp; rocprofv3 --kernel-trace --output-format csv -d /tmp/bqp -- python $GRAFT_REPO_ROOT/tools/mb_ball_query.py > /tmp/bqp.out 2>/tmp/bqp.err
cd $GRAFT_REPO_ROOT
grep grid /tmp/bqp.out
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/bqp/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "bq_" in r["Kernel_Name"]]
# group consecutive runs of 200 calls (20 per graph x 13 replays) per config: print mean per kernel in order of appearance blocks
import itertools
seq=[(r["Kernel_Name"][:28], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows]
# chunks of 2*20*14 = 560 launches per config (warm call + capture + 13 replays) -- just print block means of 520
blk=2*20*13
i=0
while i+blk<=len(seq):
    b=seq[i:i+blk]
    d=collections.defaultdict(list)
    for n,t in b: d[n].append(t)
    print({k: round(sum(v)/len(v),1) for k,v in d.items()})
    i+=blk+2*21  # eager warm call + capture pass
PY
