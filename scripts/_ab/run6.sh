cd $GRAFT_REPO_ROOT
echo "== default"; python scripts/graph_branch_probe.py 2>&1 | grep -v amdgpu.ids
echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python scripts/graph_branch_probe.py 2>&1 | grep -v amdgpu.ids
echo "== DEBUG_HIP_FORCE_GRAPH_QUEUES=4"; DEBUG_HIP_FORCE_GRAPH_QUEUES=4 python scripts/graph_branch_probe.py 2>&1 | grep -v amdgpu.ids
echo "== both"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 python scripts/graph_branch_probe.py 2>&1 | grep -v amdgpu.ids
