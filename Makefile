# Build recipe for the MI355X hot path of GSLAM.  gfx950 only.  `make` = everything buildable here.
#   lib      gslam_amd/lib/libgslam_hip.so        C-ABI + HIP kernels (the product)
#   oracle   oracle/liboracle.so                  CPU restatement (test infrastructure)
#   ref      oracle/_ref/libgslam_ref*.so         the reference's own code, compiled from /root/reference
#   plugins  gslam_amd/lib/libgslam_optimizer.so, libgslam_featuredetector.so + build/plugin_host
#            (need the GSLAM headers at BUILD time only; the .so files travel to the GPU box)
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
ARCH      ?= gfx950
REF       ?= /root/reference
# (-ffp-contract=off: the float vocabulary distance and the LM arithmetic must round like the oracle's.  Never add
#  -fgpu-flush-denormals-to-zero / -ffast-math: orb_fast_cells compares bytes held as fp16 DENORMALS with
#  v_pk_minimum3_f16 / v_pk_maximum3_f16 and relies on the default float mode that preserves them --
#  tests/test_build_float_mode.py (no GPU needed) and tests/test_orb_adversarial_gpu.py::test_arc_score_paths_agree_bit_for_bit
#  would catch it.)
HIPFLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -Iinclude
ifdef WHATIF
HIPFLAGS  += -DGH_FLOW_WHATIF $(WHATIF_FLAGS)  # timing experiments only (tools/flow_whatif.py)
endif
CFLAGS    := -O3 -fPIC -std=c11 -ffp-contract=off -fopenmp -Wall -Wno-unknown-pragmas -Iinclude
# the reference's own flags (CMakeLists.txt:9-11) for the reference shim
REFFLAGS  := -O3 -DNDEBUG -std=c++11 -fPIC -fopenmp -w -I$(REF)

CSRC      := $(wildcard gslam_amd/csrc/*.hip)
COBJ      := $(patsubst gslam_amd/csrc/%.hip,build/obj/%.o,$(CSRC))
OSRC      := $(wildcard oracle/*.c)
LIBDIR    := gslam_amd/lib

HAVE_REF  := $(wildcard $(REF)/GSLAM/core/GSLAM.h)

.PHONY: all lib oracle ref plugins refapps clean
ifeq ($(HAVE_REF),)
all: lib oracle
else
all: lib oracle ref plugins
endif

lib: $(LIBDIR)/libgslam_hip.so
oracle: oracle/liboracle.so oracle/liboracle_fma.so
ref: oracle/_ref/libgslam_ref.so oracle/_ref/libgslam_ref_popcnt.so
plugins: $(LIBDIR)/libgslam_optimizer.so $(LIBDIR)/libgslam_featuredetector.so $(LIBDIR)/libgslam_vocabulary.so $(LIBDIR)/libgslam_orbhip.so $(LIBDIR)/libgslam_estimator.so $(LIBDIR)/libgslamDB_synthplane.so $(LIBDIR)/libgslamDB_tumrgbd.so $(LIBDIR)/libgslamDB_kitti.so build/plugin_host build/gmap_check refapps

# the MFMA matcher reads its accumulators with the VALU right away: keep them in VGPRs (no v_accvgpr_read per pair)
build/obj/bf_match_mfma.o build/obj/orb.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form

build/obj/%.o: gslam_amd/csrc/%.hip gslam_amd/csrc/common.h include/gslam_hip.h $(wildcard include/*.h gslam_amd/csrc/*.h)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libgslam_hip.so: $(COBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(COBJ) -lpthread -ldl -lrt

oracle/liboracle.so: $(OSRC) $(wildcard include/*.h oracle/*.h)
	gcc $(CFLAGS) -shared -o $@ $(OSRC) -lm

# the same checker compiled WITH floating-point contraction: used only to measure how far two correct evaluations of the
# BA algorithm drift apart through rounding alone (tests/test_ba_oracle.py)
oracle/liboracle_fma.so: $(OSRC) $(wildcard include/*.h oracle/*.h)
	gcc $(subst -ffp-contract=off,-ffp-contract=fast -mfma,$(CFLAGS)) -shared -o $@ $(OSRC) -lm

oracle/_ref/libgslam_ref.so: oracle/ref_shim.cpp
	@mkdir -p oracle/_ref
	g++ $(REFFLAGS) -shared -o $@ $<

oracle/_ref/libgslam_ref_popcnt.so: oracle/ref_shim.cpp
	@mkdir -p oracle/_ref
	g++ $(REFFLAGS) -mpopcnt -shared -o $@ $<

PLUGFLAGS := -O2 -std=c++11 -fPIC -w -I$(REF) -Iinclude -Igslam_amd/plugin
$(LIBDIR)/libgslam_optimizer.so: gslam_amd/plugin/optimizer_plugin.cpp include/gslam_hip.h $(LIBDIR)/libgslam_hip.so
	g++ $(PLUGFLAGS) -shared -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN' -lpthread -ldl

$(LIBDIR)/libgslam_featuredetector.so: gslam_amd/plugin/featuredetector_plugin.cpp gslam_amd/plugin/FeatureDetector.h include/gslam_hip.h $(LIBDIR)/libgslam_hip.so
	g++ $(PLUGFLAGS) -shared -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN' -lpthread -ldl

$(LIBDIR)/libgslam_vocabulary.so: gslam_amd/plugin/vocabulary_plugin.cpp include/gslam_hip.h $(LIBDIR)/libgslam_hip.so
	g++ $(PLUGFLAGS) -shared -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN' -lpthread -ldl

$(LIBDIR)/libgslam_estimator.so: gslam_amd/plugin/estimator_plugin.cpp include/gslam_hip.h $(LIBDIR)/libgslam_hip.so
	g++ $(PLUGFLAGS) -shared -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN' -lpthread -ldl

$(LIBDIR)/libgslam_orbhip.so: gslam_amd/plugin/orbhip_app.cpp gslam_amd/plugin/FeatureDetector.h
	g++ $(PLUGFLAGS) -shared -o $@ $< -lpthread -ldl

$(LIBDIR)/libgslamDB_synthplane.so: gslam_amd/plugin/dataset_synthplane.cpp include/gslam_hip.h $(LIBDIR)/libgslam_hip.so
	g++ $(PLUGFLAGS) -shared -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN' -lpthread -ldl

# TUM-RGBD reader on the stb path (the reference's own reader needs OpenCV); host I/O only, no libgslam_hip
$(LIBDIR)/libgslamDB_tumrgbd.so: gslam_amd/plugin/dataset_tumrgbd.cpp
	@mkdir -p $(LIBDIR)
	g++ $(PLUGFLAGS) -I$(REF)/GSLAM/core -shared -o $@ $< -lpthread -ldl

# KITTI odometry reader on the stb path (the reference's own reader cannot open anything in this snapshot: see the file)
$(LIBDIR)/libgslamDB_kitti.so: gslam_amd/plugin/dataset_kitti.cpp
	@mkdir -p $(LIBDIR)
	g++ $(PLUGFLAGS) -I$(REF)/GSLAM/core -shared -o $@ $< -lpthread -ldl

# test harness: loads a .gmap with the reference's OWN MapHash (GSLAM/plugins/gmap, compiled from where it lies) and
# prints what it holds
build/gmap_check: gslam_amd/plugin/gmap_check.cpp $(wildcard $(REF)/GSLAM/plugins/gmap/Map*.cpp)
	@mkdir -p build
	g++ -O2 -std=c++11 -w -I$(REF) -I$(REF)/GSLAM/plugins/gmap -o $@ $< $(REF)/GSLAM/plugins/gmap/MapHash.cpp $(REF)/GSLAM/plugins/gmap/MapFrame.cpp $(REF)/GSLAM/plugins/gmap/MapPoint.cpp -lpthread -ldl

# The reference's OWN launcher and two of its application plugins, compiled from the sources where they lie (no copy,
# the reference's flags): `gslam` (GSLAM/gslam/main.cpp), `play` (plugins/play/main.cpp), `metric_time`
# (evaluation/metric_time/main.cpp), `metric_traj` (evaluation/metric_trajectory/main.cpp).  Test infrastructure: tests/test_launcher_gpu.py drives the orbhip application and
# the synthplane dataset through them.  build/ is git-ignored and travels to the GPU box.
refapps: build/ref/gslam build/ref/libgslam_play.so build/ref/libgslam_metric_time.so build/ref/libgslam_metric_traj.so build/ref/libgslam_gmap.so
# the reference's own gmap application plugin
build/ref/libgslam_gmap.so: $(wildcard $(REF)/GSLAM/plugins/gmap/*.cpp)
	@mkdir -p build/ref
	g++ -O3 -DNDEBUG -std=c++11 -w -fPIC -shared -I$(REF) -I$(REF)/GSLAM/plugins/gmap -o $@ $^ -lpthread -ldl
build/ref/gslam: $(REF)/GSLAM/gslam/main.cpp
	@mkdir -p build/ref
	g++ -O3 -DNDEBUG -std=c++11 -w -I$(REF) -I$(REF)/GSLAM/core -o $@ $< -lpthread -ldl
build/ref/libgslam_play.so: $(REF)/GSLAM/plugins/play/main.cpp
	@mkdir -p build/ref
	g++ -O3 -DNDEBUG -std=c++11 -w -fPIC -shared -I$(REF) -o $@ $< -lpthread -ldl
build/ref/libgslam_metric_time.so: $(REF)/GSLAM/evaluation/metric_time/main.cpp
	@mkdir -p build/ref
	g++ -O3 -DNDEBUG -std=c++11 -w -fPIC -shared -I$(REF) -o $@ $< -lpthread -ldl
build/ref/libgslam_metric_traj.so: $(REF)/GSLAM/evaluation/metric_trajectory/main.cpp
	@mkdir -p build/ref
	g++ -O3 -DNDEBUG -std=c++11 -w -fPIC -shared -I$(REF) -o $@ $< -lpthread -ldl

build/plugin_host: gslam_amd/plugin/plugin_host.cpp gslam_amd/plugin/FeatureDetector.h gslam_amd/plugin/UndistorterHIP.h $(LIBDIR)/libgslam_hip.so
	@mkdir -p build
	g++ $(PLUGFLAGS) -rdynamic -o $@ $< -L$(LIBDIR) -lgslam_hip -Wl,-rpath,'$$ORIGIN/../gslam_amd/lib' -lpthread -ldl

clean:
	rm -rf build $(LIBDIR)/*.so oracle/liboracle.so oracle/liboracle_fma.so oracle/_ref
