# scratch: the command file of the last experiment run through gpurun (scripts/exp_build.sh / exp_file.sh build the libraries under
# ml-quant_amd/lib_exp/, LSQ_HIP_LIB selects one)
