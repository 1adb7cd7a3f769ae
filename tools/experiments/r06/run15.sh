cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
bash tools/profile_round.sh r06 2>&1 | tail -5
