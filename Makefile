# Build of libjpeg_amd.so (HIP kernels + C-ABI) for MI355X / gfx950, and of the test oracle.
#   make            -> jpeg_amd/libjpeg_amd.so
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
# -ffp-contract=off is REQUIRED: the float pipeline must not form FMAs (bit-exact parity).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC      = jpeg_amd/csrc
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Iinclude
SRCS      = $(CSRC)/api.hip $(CSRC)/pixels.hip $(CSRC)/huffdec.hip $(CSRC)/huffsync.hip $(CSRC)/huffenc.hip
OBJS      = $(SRCS:.hip=.o) $(CSRC)/codestream.o
HDRS      = $(CSRC)/common.hpp $(CSRC)/kernels.hpp $(CSRC)/codestream.hpp $(CSRC)/huffcore.hpp $(CSRC)/dct_lut.inc include/jpeg_amd.h

.PHONY: all oracle clean
all: jpeg_amd/libjpeg_amd.so

jpeg_amd/libjpeg_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/codestream.o: $(CSRC)/codestream.cpp $(HDRS)
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(OBJS) jpeg_amd/libjpeg_amd.so
