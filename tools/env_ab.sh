# A/B of ONE build under two environments on one box: tools/env_ab.sh "VAR=1" "<command>"  -> runs <command> alternately without / with VAR, 3 times
for rep in 1 2 3; do
  echo "== baseline (rep $rep)"; bash -c "$2" 2>&1 | grep -v amdgpu.ids
  echo "== $1 (rep $rep)"; env $1 bash -c "$2" 2>&1 | grep -v amdgpu.ids
done
