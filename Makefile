# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#
#   make            -> streamly-lz4_amd/lib/libmi355lz4.so  + oracle/
#   make lib        -> product library only
#   make lib-exp    -> the same with the shelved experiments compiled in (tests only)
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
PKG   := streamly-lz4_amd
CSRC  := $(PKG)/csrc
LIB   := $(PKG)/lib/libmi355lz4.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result

SRCS := $(CSRC)/kernels.hip $(CSRC)/api.cpp $(CSRC)/host_stream.cpp $(CSRC)/lz4_frame.cpp $(CSRC)/multi_device.cpp
HDRS := $(wildcard $(CSRC)/*.hpp $(CSRC)/*.h include/*.h)

all: lib oracle

lib: $(LIB)

$(LIB): $(SRCS) $(HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-Bsymbolic -o $@ -x hip $(SRCS)

# The library with the measured-and-shelved experiments compiled in (decoder variant 3: the token-list parse of round 4).
# Not what ships: `make lib` leaves them out.  tests/test_experiment_build_gpu.py runs the decoder parity tests on it.
LIBEXP := $(PKG)/lib/libmi355lz4_exp.so
lib-exp: $(LIBEXP)

$(LIBEXP): $(SRCS) $(HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -DMI355LZ4_EXPERIMENTS -shared -Wl,-Bsymbolic -o $@ -x hip $(SRCS)

oracle:
	$(MAKE) -C oracle

# Host-side sanitizer builds (CPU only; GPU ASan is not available on this pool): api.cpp + host_stream.cpp
# compiled by g++ against the HIP host API, kernel launchers stubbed, driven by tests/native/host_san_test.cpp.
SAN_SRCS := $(CSRC)/api.cpp $(CSRC)/host_stream.cpp $(CSRC)/lz4_frame.cpp $(CSRC)/multi_device.cpp tests/native/san_stubs.cpp tests/native/host_san_test.cpp
SAN_FLAGS := -std=c++17 -O1 -g -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wall -Wno-unused-function -Wno-unused-result
SAN_LIBS := -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lpthread

build/san/host_asan: $(SAN_SRCS) $(HDRS)
	@mkdir -p build/san
	g++ $(SAN_FLAGS) -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $@ $(SAN_SRCS) $(SAN_LIBS)

build/san/host_tsan: $(SAN_SRCS) $(HDRS)
	@mkdir -p build/san
	g++ $(SAN_FLAGS) -fsanitize=thread -o $@ $(SAN_SRCS) $(SAN_LIBS)

asan: build/san/host_asan
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 LSAN_OPTIONS=suppressions=tests/native/lsan.supp ./build/san/host_asan

tsan: build/san/host_tsan
	TSAN_OPTIONS=halt_on_error=1 ./build/san/host_tsan

clean:
	rm -rf build/san
	rm -f $(LIB) $(LIBEXP)
	$(MAKE) -C oracle clean

.PHONY: all lib lib-exp oracle clean asan tsan
