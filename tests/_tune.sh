for T in "1,1,1,1,1,1,1,1000" "1,1,1,1,1,1,1,16" "2,1,1,2,2,1,1,16" "2,1,1,2,2,1,1,24"; do
  echo "TUNE $T: $(RTOW_TUNE=$T python tests/run_gpu_quick.py 1920 1080 64 8 2>&1 | grep 'iter 2')"
done
