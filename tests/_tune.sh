for T in "1,1,1,1,1,1,1,16" "1,1,1,1,1,1,1,16" "1,1,1,1,1,1,1,20"; do
  echo "TUNE $T: $(RTOW_TUNE=$T python tests/run_gpu_quick.py 1920 1080 64 8 2>&1 | grep 'iter 2')"
done
