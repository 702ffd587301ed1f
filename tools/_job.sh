cd $GRAFT_REPO_ROOT; bash tools/trace_job.sh
