# In-kernel shader clock under the conv kernels and the matrix-core ceiling of this box (scripts/mfma_peak.hip,
# scripts/convlab.hip built with -DPN_STAMP); run from the repo root on the GPU box.
cd popnet_amd/build
for w in 4 2 1; do timeout 60 ./mfma_peak $w 2 1; done
timeout 60 ./mfma_peak 4 2 0
export NBUF=1
echo "== stage level (256->256 + 128->128 + 128->64, 28x28, B=32), 50000 back-to-back launches"
GROUP="128:128,128:64" timeout 120 ./convlab_stamp 32 28 28 256 256 3 0 50000 v3 0 | grep "us/launch\|clock\|block starts"
echo "== 256->256 alone"
timeout 120 ./convlab_stamp 32 28 28 256 256 3 0 50000 v3 0 | grep "us/launch\|clock\|block starts"
echo "== 112x112 64->64 with residual"
timeout 120 ./convlab_stamp 32 112 112 64 64 3 1 30000 v3 1 | grep "us/launch\|clock\|block starts"
