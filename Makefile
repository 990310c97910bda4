# Builds libhbegp.so (HIP kernels + host runtime + C ABI) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CXXFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result
LIB = hbetune_rs_amd/libhbegp.so
SRC = csrc/kernels.hip csrc/hbegp.cpp
HDR = csrc/engine.hpp csrc/lbfgsb.hpp csrc/lbfgs_step.hpp csrc/dag_plan.hpp csrc/dag_kernel.inc.hpp csrc/fastmath.hpp include/hbegp.h

all: $(LIB)

build/kernels.o: csrc/kernels.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@
build/hbegp.o: csrc/hbegp.cpp $(HDR)
	@mkdir -p build
	$(HIPCC) -O3 -std=c++17 -fPIC -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -x c++ -c $< -o $@
# device ISA of the kernels (tests/test_isa_guards_cpu.py: DPP hazard scan + spill counts of the default instantiations)
build/kernels.s: csrc/kernels.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) --cuda-device-only -S $< -o $@
$(LIB): build/kernels.o build/hbegp.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread

# experimental builds of the kernels (diagnostics only; loaded with HBEGP_LIB=build/var/libhbegp_<NAME>.so):
#   make variant NAME=noinline DEFS="-DDAG_LEAF_NOINLINE=1"
variant: build/hbegp.o
	@mkdir -p build/var
	$(HIPCC) $(CXXFLAGS) $(DEFS) -c csrc/kernels.hip -o build/var/kernels_$(NAME).o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/var/libhbegp_$(NAME).so build/var/kernels_$(NAME).o build/hbegp.o -lpthread

oracle: oracle/libgpr_oracle.so oracle/libreferee.so
oracle/libreferee.so: oracle/referee.c
	gcc -O2 -fPIC -shared -fopenmp -mfma -ffp-contract=off -o $@ $< -lquadmath -lm
oracle/libgpr_oracle.so: oracle/gpr_oracle.c
	gcc -O2 -fPIC -shared -o $@ $< -lm

clean:
	rm -rf build $(LIB) oracle/libgpr_oracle.so oracle/libreferee.so
.PHONY: all clean oracle variant
