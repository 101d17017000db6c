for NF in 3 4 6 8; do for SZ in "" "--height 480 --width 640"; do
FDT_HOST_NF=$NF python bench.py --steps 32 --warmup 8 --cpu-frames 0 --host-frames 192 --profile-frames 1 $SZ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NF', '$SZ', d['value'], d['host_path']['value'])"
done; done
