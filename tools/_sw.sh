for v in DVG_SKIP_HOIST=0 DVG_UPCONV_AS_CONVT=0 DVG_WINOGRAD=0 DVG_WINOGRAD_CHAIN=0 DVG_FIRST_PAIR=0 DVG_NO_SPLITK=1 DVG_SPLITK_ONE_LAUNCH=1 DVG_UPCONV_WINOGRAD=0 DVG_LATENT_STREAM=0; do
  echo "== $v (tests/test_gpu_rollouts.py tests/test_gpu_generate_config.py tests/test_gpu_determinism.py)"
  env $v timeout 600 python -m pytest tests/test_gpu_rollouts.py tests/test_gpu_generate_config.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -1
done 2>&1 | tee gpurun_out/sw_rollouts.txt
