cd "$GRAFT_REPO_ROOT"
for lib in build_ab/libdnaf_gtz_exp*.so; do n=$(basename $lib .so)
  DNAF_LIB=$PWD/$lib python tools/bgzf_experiments/gtz_bymaf.py 2>/dev/null | awk '/^p=/{printf "%s ", $(NF-2)} END{print ""}' | sed "s/^/$n by class ms: /"
done
