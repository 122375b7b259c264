export FIESTA_REV=307c0c1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/collect_evidence.sh r03f 2>&1 | tail -5
ls gpurun_out/r03f
