cd $GRAFT_REPO_ROOT
for p in tools/_probe/p_*; do echo "== $p"; timeout 60 $p 1024 64 0; done 2>&1 | tee gpurun_out/r5_skinny_probe2.txt
