# Plain-make build of the product for C/C++ users (python -m libsrcnn_amd.build does the same).
#   make            -> libsrcnn_amd/lib/libsrcnn_amd.so + libsrcnn_amd/bin/srcnntest
#   make oracle     -> the CPU checker (and oracle/_ref where the reference tree is present)
#   make test       -> CPU test-suite;  make gpu-test on a gfx950 box
HIPCC   ?= /opt/rocm/bin/hipcc
CSRC    := libsrcnn_amd/csrc
LIBDIR  := libsrcnn_amd/lib
BINDIR  := libsrcnn_amd/bin
# -ffp-contract=off: the strict kernels and the host table builder must round every multiply and add separately
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wall \
            -Wno-unused-result -Wno-unused-value -Wno-ignored-attributes -D__HIP_PLATFORM_AMD__
SRCS    := srcnn_kernels.hip srcnn_capi.cpp srcnn_comm.cpp dropin.cpp
OBJS    := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(basename $(SRCS))))
HDRS    := $(CSRC)/srcnn_kernels.h $(CSRC)/resample_table.hpp $(CSRC)/srcnn_weights.inc include/srcnn_amd.h include/libsrcnn_dropin.h

all: $(LIBDIR)/libsrcnn_amd.so $(BINDIR)/srcnntest

$(LIBDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/libsrcnn_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJS) -o $@ -ldl

$(BINDIR)/srcnntest: tools/srcnntest.cpp $(LIBDIR)/libsrcnn_amd.so
	@mkdir -p $(BINDIR)
	g++ -O2 -std=c++17 $< -L$(LIBDIR) -lsrcnn_amd -Wl,-rpath,$(abspath $(LIBDIR)) -Wl,-rpath,'$$ORIGIN/../lib' -o $@

oracle:
	$(MAKE) -C oracle all
	@if [ -d /root/reference/src ]; then $(MAKE) -C oracle ref; fi

test: all oracle
	python -m pytest tests -x -q -m "not gpu"

gpu-test: all oracle
	python -m pytest tests -x -q -m gpu

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/_build oracle/_ref

.PHONY: all oracle test gpu-test clean
