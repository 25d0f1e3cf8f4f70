# round 5: cache-policy variants of the batched K1 (tools/experiments/libpca_<v>.bin built with -DK1_STREAM_NT / -DK1_GATHER_MODE),
# each timed on 64 uniform and 64 ring-model frames, with k1_append plain and non-temporal
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
for v in ${VARIANTS:-std s1 s2 s3 g1 g2 g3 g4 s3g1}; do
  cp tools/experiments/libpca_$v.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  for KIND in uniform ring; do
    ARGS="k1 64"; [ $KIND = ring ] && ARGS="k1 64 ring"
    for A in 1 1,nt; do
      echo "$v $KIND append=$A: $(PCA_K1_APPEND=$A python tools/experiments/pass_only.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wall us %.2f  events us %.2f  frac %.4f' % (d['us_per_call_wall_back_to_back'], d['us_per_call_hip_events'], d['frac']))")"
    done
  done
done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
