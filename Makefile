# Builds librtamd.so (host C++ + gfx950 HIP kernels), the headless driver, and the test oracle.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
CS = raytrace_amd/csrc
OUT = raytrace_amd/librtamd.so
BUILD = build

# Host code keeps every float operation unfused and in source order; the kernels get the
# same setting so the triangle test reproduces the CPU arithmetic bit for bit, and divides
# stay correctly rounded.
HOSTFLAGS = -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude
HIPFLAGS = -std=c++17 -O3 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
           -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-parameter -Iinclude $(EXTRA_HIPFLAGS)

HOST_SRCS = obj_loader centroid_bvh sah_bvh raygen flatten framebuffer
HIP_SRCS = trace_kernels flatten_device capi multi_gpu
HOST_OBJS = $(foreach s,$(HOST_SRCS),$(BUILD)/$(s).o)
HIP_OBJS = $(foreach s,$(HIP_SRCS),$(BUILD)/$(s).o)
HDRS = $(wildcard $(CS)/*.h) $(wildcard include/*.h)

all: $(OUT) tools/rt_headless oracle

$(BUILD)/%.o: $(CS)/%.cpp $(HDRS)
	@mkdir -p $(BUILD)
	$(CXX) $(HOSTFLAGS) -c $< -o $@

$(BUILD)/%.o: $(CS)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(HOST_OBJS) $(HIP_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -Wl,-rpath,$(ROCM)/lib -lpthread -ldl

tools/rt_headless: tools/rt_headless.cpp $(OUT) $(HDRS)
	$(CXX) -std=c++17 -O2 -Iinclude -I$(ROCM)/include -D__HIP_PLATFORM_AMD__ -o $@ tools/rt_headless.cpp -Lraytrace_amd -lrtamd -L$(ROCM)/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../raytrace_amd' -Wl,-rpath,$(ROCM)/lib

oracle:
	$(MAKE) -C oracle

# The host half of the library (no GPU code) under AddressSanitizer + UndefinedBehaviorSanitizer.
sanitize-host: $(BUILD)/host_selftest_san
	ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $(BUILD)/host_selftest_san assets/models

$(BUILD)/host_selftest_san: tools/host_selftest.cpp $(foreach s,$(HOST_SRCS),$(CS)/$(s).cpp) $(HDRS)
	@mkdir -p $(BUILD)
	$(CXX) -std=c++17 -O1 -g -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
	    -o $@ tools/host_selftest.cpp $(foreach s,$(HOST_SRCS),$(CS)/$(s).cpp) -lpthread

clean:
	rm -rf $(BUILD) $(OUT) tools/rt_headless
	$(MAKE) -C oracle clean

.PHONY: all oracle clean sanitize-host
