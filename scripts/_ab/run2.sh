cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a
bash scripts/prof_step.sh r05a/fused
python scripts/prof_by_geometry.py gpurun_out/r05a/fused/p_results.db 13 4 > gpurun_out/r05a/fused_geom.txt
SAUNET_DENSE_BWD_FUSED=0 bash scripts/prof_step.sh r05a/unfused
python scripts/prof_by_geometry.py gpurun_out/r05a/unfused/p_results.db 13 4 > gpurun_out/r05a/unfused_geom.txt
rm -rf gpurun_out/r05a/fused gpurun_out/r05a/unfused
head -3 gpurun_out/r05a/fused_geom.txt gpurun_out/r05a/unfused_geom.txt
