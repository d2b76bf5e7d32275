# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#
#   make            -> streamly-lz4_amd/lib/libmi355lz4.so  + oracle/
#   make lib        -> product library only
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
PKG   := streamly-lz4_amd
CSRC  := $(PKG)/csrc
LIB   := $(PKG)/lib/libmi355lz4.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result

SRCS := $(CSRC)/kernels.hip $(CSRC)/api.cpp $(CSRC)/host_stream.cpp
HDRS := $(wildcard $(CSRC)/*.hpp $(CSRC)/*.h include/*.h)

all: lib oracle

lib: $(LIB)

$(LIB): $(SRCS) $(HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-Bsymbolic -o $@ -x hip $(SRCS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
