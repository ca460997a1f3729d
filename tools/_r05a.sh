export T_SEL="tests/test_conv.py::test_wgrad_stride2_row_sharing tests/test_conv.py::test_forward_statistics_with_a_residual_use_the_full_plans_row_count tests/test_conv.py::test_stream_kernel_statistics_rows_follow_the_grid tests/test_conv.py::test_bench_instantiations_elementwise tests/test_ssod_step.py tests/test_parallel.py tests/test_nms.py tests/test_abi.py tests/test_norm_spatial.py"
export AB_ENV="X=0 ET_TEACHER_CUS=64 ET_TEACHER_CUS=128 X=0 ET_TEACHER_CUS=96 ET_TEACHER_CUS=64/4 ET_WGRAD_STREAM=1 ET_TEACHER_CUS=192"
bash tools/r05_gpu.sh r05a env pmc_list t_sel bench pmc_mfma ab_env
