# mm_discrete_split_kernel with other noise-wave counts / ring depths: builds tools/experiments/ds/lib_NN_RB.so (here, CPU)
# usage: bash tools/experiments/ds_knobs.sh "3 12" "3 24" ...     then on the GPU box: python tools/experiments/ds_knobs.py
cd "$(dirname "$0")/../../mini_mcmc_amd/csrc" && mkdir -p ../../tools/experiments/ds
for v in "$@"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DMM_DS_NN=$1 -DMM_DS_RB=$2 $DS_EXTRA -c mm_discrete.hip -o /tmp/ds_$1_$2.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/experiments/ds/lib_$1_$2.so $(ls build/*.o | grep -v mm_discrete.o) /tmp/ds_$1_$2.o -ldl -lpthread
done
