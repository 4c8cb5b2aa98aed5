echo "240 steps: base | DIV=1 | DIV=3 | HANDOFF=2 | ROWS=16 | ROWS=30 | ROWS_NEW=4 | ROWS_NEW=10"
bash tools/ab_mix.sh "--steps 240 --warmup 3" "cur.so" "cur.so ROFT_OUTLIER_STEADY_DIV=1" "cur.so ROFT_OUTLIER_STEADY_DIV=3" "cur.so ROFT_HANDOFF=2" "cur.so ROFT_MASK_ROWS=16" "cur.so ROFT_MASK_ROWS=30" "cur.so ROFT_MASK_ROWS_NEW=4" "cur.so ROFT_MASK_ROWS_NEW=10"
