for e in "X=1" "MFVIT_NT_DEEP=0" "MFVIT_NT_STORE=0" "MFVIT_ROW_LEAN=0"; do
echo "$e: $(env $e python bench.py --workload single --batch 64 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c80-200)"
done
git -C . log --oneline | head -3
