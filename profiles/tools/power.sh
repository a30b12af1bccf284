for v in product; do
  for mode in 3 1 0; do
    ./profiles/tools/gloop protein_transformer_amd/csrc/libptamd.so $mode 4096 4096 4096 6 &
    PID=$!
    sleep 2.5
    for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.8; done
    wait $PID
  done
done
