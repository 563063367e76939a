for s in "8192 262144" "65536 131072" "131072 262144" "131072 1250000"; do set -- $s
bash tools/ab_lib.sh 2 python3 tools/route_ab.py --users $1 --items $2 --arms default
done
