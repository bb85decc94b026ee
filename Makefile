# Plain-make build of the product for C/C++ users (python -m libsrcnn_amd.build does the same).
#   make            -> libsrcnn_amd/lib/libsrcnn_amd.so + libsrcnn_amd/bin/srcnntest
#   make oracle     -> the CPU checker (and oracle/_ref where the reference tree is present)
#   make test       -> CPU test-suite;  make gpu-test on a gfx950 box
#   make ubench     -> tools/ubench/bin/* (microbenchmarks; hipcc, gfx950)
#   make asan tsan  -> the host code (table builder, drop-in control flow, oracle) under ASan+UBSan / TSan, CPU only
HIPCC   ?= /opt/rocm/bin/hipcc
CSRC    := libsrcnn_amd/csrc
LIBDIR  := libsrcnn_amd/lib
BINDIR  := libsrcnn_amd/bin
# -ffp-contract=off: the strict kernels and the host table builder must round every multiply and add separately
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wall \
            -Wno-unused-result -Wno-unused-value -Wno-ignored-attributes -D__HIP_PLATFORM_AMD__
SRCS    := srcnn_kernels.hip srcnn_fused_f16.hip srcnn_capi.cpp srcnn_pipeline.cpp srcnn_comm.cpp dropin.cpp
OBJS    := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(basename $(SRCS))))
HDRS    := $(CSRC)/srcnn_kernels.h $(CSRC)/srcnn_host.hpp $(CSRC)/resample_table.hpp $(CSRC)/srcnn_weights.inc include/srcnn_amd.h include/libsrcnn_dropin.h

all: $(LIBDIR)/libsrcnn_amd.so $(BINDIR)/srcnntest

$(LIBDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -x hip -c $< -o $@
# the fused kernel is issue-bound beside its MFMAs, where packed fp32 VALU ops are an anti-lever: no SLP packing there
EXTRA_srcnn_fused_f16 := -fno-slp-vectorize

$(LIBDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/libsrcnn_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJS) -o $@ -ldl

$(BINDIR)/srcnntest: tools/srcnntest.cpp $(LIBDIR)/libsrcnn_amd.so
	@mkdir -p $(BINDIR)
	g++ -O2 -std=c++17 $< -L$(LIBDIR) -lsrcnn_amd -Wl,-rpath,$(abspath $(LIBDIR)) -Wl,-rpath,'$$ORIGIN/../lib' -o $@

# the microbenchmarks behind profiles/r0x_*.txt (outputs are NOT tracked: tools/ubench/bin/ is git-ignored)
UBENCH_HIP := $(wildcard tools/ubench/*.hip) $(wildcard tools/ubench/*.cpp)
ubench: $(patsubst tools/ubench/%,tools/ubench/bin/%,$(basename $(UBENCH_HIP)))
tools/ubench/bin/%: tools/ubench/%.hip
	@mkdir -p tools/ubench/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $< -o $@
tools/ubench/bin/%: tools/ubench/%.cpp
	@mkdir -p tools/ubench/bin
	$(HIPCC) --offload-arch=gfx950 -O2 -std=c++17 -x hip $< -o $@ -lpthread

oracle:
	$(MAKE) -C oracle all
	@if [ -d /root/reference/src ]; then $(MAKE) -C oracle ref; fi

test: all oracle
	python -m pytest tests -x -q -m "not gpu"

gpu-test: all oracle
	python -m pytest tests -x -q -m gpu

# CPU-only sanitizer builds of the host code (no GPU sanitizer exists on this pool): tests/host/host_sanitize.cpp
SAN_SRCS := tests/host/host_sanitize.cpp $(CSRC)/dropin.cpp
SAN_DEPS := $(SAN_SRCS) $(CSRC)/resample_table.hpp oracle/srcnn_oracle.c include/srcnn_amd.h include/libsrcnn_dropin.h
tests/host/_build/oracle_%.o: oracle/srcnn_oracle.c oracle/oracle_weights.inc
	@mkdir -p tests/host/_build
	gcc -O1 -g -ffp-contract=off -std=c99 -fsanitize=$(subst asan,address$(comma)undefined,$(subst tsan,thread,$*)) -fno-omit-frame-pointer -c $< -o $@
comma := ,
tests/host/_build/host_asan: $(SAN_DEPS) tests/host/_build/oracle_asan.o
	g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
	    $(SAN_SRCS) tests/host/_build/oracle_asan.o -lm -o $@
tests/host/_build/host_tsan: $(SAN_DEPS) tests/host/_build/oracle_tsan.o
	g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=thread -fno-omit-frame-pointer \
	    $(SAN_SRCS) tests/host/_build/oracle_tsan.o -lm -lpthread -o $@
asan: tests/host/_build/host_asan
	ASAN_OPTIONS=detect_leaks=1:alloc_dealloc_mismatch=1:strict_string_checks=1 UBSAN_OPTIONS=print_stacktrace=1 $<
tsan: tests/host/_build/host_tsan
	TSAN_OPTIONS=halt_on_error=1 $<

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/_build oracle/_ref tests/host/_build tools/ubench/bin

.PHONY: all oracle test gpu-test clean asan tsan ubench
