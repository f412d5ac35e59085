# Two host threads, two contexts (tools/ab_two_threads.py) under the HIP runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4):
#   bash tools/ab_hw_queues.sh
cd $GRAFT_REPO_ROOT
for q in 4 8 6 8 4 16; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python3 tools/ab_two_threads.py 2>/dev/null
done
