export FIESTA_REV=5d221c3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/collect_evidence.sh r03g 2>&1 | tail -3
