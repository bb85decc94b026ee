# Plain-make build of the product for C/C++ users (python -m libsrcnn_amd.build does the same).
#   make            -> libsrcnn_amd/lib/libsrcnn_amd.so + libsrcnn.so + libsrcnn.a + libsrcnn_amd/bin/srcnntest
#                      libsrcnn.so / libsrcnn.a: the names the reference's Makefiles produce (Makefiles/Makefile.linux:13-14,
#                      38-39) -- a program linked with -lsrcnn runs on this library without relinking
#   make install    -> $(DESTDIR)$(PREFIX)/lib/{libsrcnn_amd.so,libsrcnn.so,libsrcnn.a}, include/{libsrcnn.h,srcnn_amd.h}
#                      (the reference's install / uninstall: Makefiles/Makefile.linux:64-75)
#   make oracle     -> the CPU checker (and oracle/_ref where the reference tree is present)
#   make test       -> CPU test-suite;  make gpu-test on a gfx950 box
#   make ubench     -> tools/ubench/bin/* (microbenchmarks; hipcc, gfx950)
#   make asan tsan  -> the host code (table builder, drop-in control flow, oracle) under ASan+UBSan / TSan, CPU only
#   make STRICT_ONLY=1 -> the same artefacts under libsrcnn_amd/lib/strict/ with NO non-parity kernel compiled in (no FAST /
#                      FAST_F16 / RELAXED template instance, no srcnn_fused_f16.hip); srcnn_set_mode refuses those modes
HIPCC   ?= /opt/rocm/bin/hipcc
CSRC    := libsrcnn_amd/csrc
ifeq ($(STRICT_ONLY),1)
LIBDIR  := libsrcnn_amd/lib/strict
BINDIR  := libsrcnn_amd/lib/strict/bin
STRICT_DEF := -DSRCNN_STRICT_ONLY
else
LIBDIR  := libsrcnn_amd/lib
BINDIR  := libsrcnn_amd/bin
STRICT_DEF :=
endif
# -ffp-contract=off: the strict kernels and the host table builder must round every multiply and add separately
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wall \
            -Wno-unused-result -Wno-unused-value -Wno-ignored-attributes -D__HIP_PLATFORM_AMD__ $(STRICT_DEF)
ifeq ($(STRICT_ONLY),1)
SRCS    := srcnn_kernels.hip srcnn_capi.cpp srcnn_pipeline.cpp srcnn_comm.cpp dropin.cpp
else
SRCS    := srcnn_kernels.hip srcnn_fused_f16.hip srcnn_capi.cpp srcnn_pipeline.cpp srcnn_comm.cpp dropin.cpp
endif
OBJS    := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(basename $(SRCS))))
HDRS    := $(CSRC)/srcnn_kernels.h $(CSRC)/srcnn_host.hpp $(CSRC)/srcnn_settings.hpp $(CSRC)/srcnn_watchdog.hpp $(CSRC)/resample_table.hpp $(CSRC)/srcnn_weights.inc include/srcnn_amd.h include/srcnn_amd_debug.h include/libsrcnn_dropin.h

PREFIX  ?= /usr/local
ROCM    ?= /opt/rocm

all: $(LIBDIR)/libsrcnn_amd.so $(LIBDIR)/libsrcnn.so $(LIBDIR)/libsrcnn.a $(BINDIR)/srcnntest

$(LIBDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -x hip -c $< -o $@
# the fused kernel is issue-bound beside its MFMAs, where packed fp32 VALU ops are an anti-lever: no SLP packing there
EXTRA_srcnn_fused_f16 := -fno-slp-vectorize

$(LIBDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# exports.map: only srcnn_* and the two reference symbols are exported (the kernels' launch stubs are not)
$(LIBDIR)/libsrcnn_amd.so: $(OBJS) $(CSRC)/exports.map
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJS) -o $@ -ldl -Wl,-soname,libsrcnn_amd.so -Wl,--version-script=$(CSRC)/exports.map

# libsrcnn.so: what a program linked against the reference's shared library asks the loader for (the reference links it
# without a soname, so the request is the file name): a symbolic link to the product.  New links with -lsrcnn record the
# product's soname (libsrcnn_amd.so), old binaries find the file they ask for.
$(LIBDIR)/libsrcnn.so: $(LIBDIR)/libsrcnn_amd.so
	ln -sf libsrcnn_amd.so $@

# libsrcnn.a: the reference's DEFAULT artefact is the static library.  Consumers add the HIP runtime to their link line:
#   g++ app.o -L<libdir> -lsrcnn -L$(ROCM)/lib -lamdhip64 -ldl -lpthread      (instead of the reference's -lsrcnn -fopenmp)
$(LIBDIR)/libsrcnn.a: $(OBJS)
	rm -f $@ && ar crs $@ $(OBJS)

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
SAN_DEPS := $(SAN_SRCS) $(CSRC)/resample_table.hpp $(CSRC)/srcnn_watchdog.hpp oracle/srcnn_oracle.c include/srcnn_amd.h include/srcnn_amd_debug.h include/libsrcnn_dropin.h
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

install: all
	install -d $(DESTDIR)$(PREFIX)/lib $(DESTDIR)$(PREFIX)/include
	install -m 755 $(LIBDIR)/libsrcnn_amd.so $(DESTDIR)$(PREFIX)/lib/
	ln -sf libsrcnn_amd.so $(DESTDIR)$(PREFIX)/lib/libsrcnn.so
	install -m 644 $(LIBDIR)/libsrcnn.a $(DESTDIR)$(PREFIX)/lib/
	install -m 644 include/libsrcnn_dropin.h $(DESTDIR)$(PREFIX)/include/libsrcnn.h
	install -m 644 include/srcnn_amd.h $(DESTDIR)$(PREFIX)/include/srcnn_amd.h
	@if [ -z "$(DESTDIR)" ] && [ "$$(id -u)" = 0 ]; then ldconfig; fi

uninstall:
	rm -f $(DESTDIR)$(PREFIX)/lib/libsrcnn_amd.so $(DESTDIR)$(PREFIX)/lib/libsrcnn.so $(DESTDIR)$(PREFIX)/lib/libsrcnn.a
	rm -f $(DESTDIR)$(PREFIX)/include/libsrcnn.h $(DESTDIR)$(PREFIX)/include/srcnn_amd.h
	@if [ -z "$(DESTDIR)" ] && [ "$$(id -u)" = 0 ]; then ldconfig; fi

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/_build oracle/_ref tests/host/_build tools/ubench/bin

.PHONY: all oracle test gpu-test clean asan tsan ubench install uninstall
