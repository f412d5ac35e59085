# A/B of two builds of the library on the SAME box (VALU-bound kernels differ by several % between boxes):
#   bash tools/ab_chain.sh stripenn_amd/libstp_ab_old.so stripenn_amd/libstripenn_hip.so
R=$(pwd)
for rep in 1 2; do
  for l in "$@"; do
    echo "$l: $(STP_LIB=$R/$l python3 tools/probe_chain.py | tail -1)"
  done
done
