# tools/variant_run.sh "<command>" : the command under the in-tree build and under every ab_variants/lib_*.so (ALVA_LIB), twice round-robin
for rep in 1 2; do
  for lib in "" ab_variants/lib_*.so; do
    echo "== ${lib:-in-tree} (rep $rep)"
    ALVA_LIB=$lib bash -c "$1" 2>&1 | grep -v amdgpu.ids
  done
done
