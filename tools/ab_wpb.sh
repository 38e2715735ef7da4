export MC_JIT_CACHE=/tmp/jc
for c in "4 4" "4 8" "4 16" "8 4" "8 8" "4 4"; do
  set -- $c
  export MC_WPB_CLASSIFY=$1 MC_WPB_EMIT=$2
  timeout -k 10 120 python tools/ab.py "c$1e$2=" 2>&1 | grep rep2
done
