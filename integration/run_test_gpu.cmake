# GPU twin of the reference's test/run_test.cmake:1-10: run the codec application, then print the md5s of the syntax-
# element log and of the output for the ctest's PASS_REGULAR_EXPRESSION.  Differences: it reports "skipped: ..." (ctest:
# SKIP_REGULAR_EXPRESSION) instead of failing when the application or a GPU is not there, and it makes the library's
# "no CPU fallback" visible by refusing to run with the GPU path switched off.
#   cmake -DCMD=<EncoderApp|DecoderApp> -DARGS="..." -DLOG_FILE=bin_log.txt -DOUT_FILE=<str.bin|recon.yuv> -P run_test_gpu.cmake
if(NOT CMD OR NOT EXISTS "${CMD}")
  message("skipped: VTM/clips not supplied - ${CMD} does not exist")
  return()
endif()
if(NOT EXISTS "/dev/kfd")
  message("skipped: no AMD GPU (/dev/kfd) on this machine; libcabac_hip.so has no CPU path")
  return()
endif()

set(ARGS_LIST ${ARGS})
separate_arguments(ARGS_LIST)

execute_process(COMMAND ${CMD} ${ARGS_LIST} RESULT_VARIABLE CMD_RESULT COMMAND_ECHO STDOUT)
if(CMD_RESULT)
  message(FATAL_ERROR "Error running ${CMD}: ${CMD_RESULT}")
endif()

execute_process(COMMAND md5sum ${LOG_FILE})
execute_process(COMMAND md5sum ${OUT_FILE})
