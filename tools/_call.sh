cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/ab.sh r3x17 "base:" "rec3:RICADI_RECYCLE=3" "rec8:RICADI_RECYCLE=8" "cyc8:RICADI_CYC0=8" "cyc12:RICADI_CYC0=12" "cyc15:RICADI_CYC0=15" "ap20:RICADI_OPTS=agg_p=20" "ap32x:RICADI_OPTS=agg_p=36" "av12:RICADI_OPTS=agg_v=12" "av20:RICADI_OPTS=agg_v=20" "base2:"
