/* TEST-ONLY entry points of libalvaar_hip.so -- not part of the drop-in surface (include/alvaar_system.h); a caller that replaces the
 * reference's `System` never needs this header.  Used by tests/ (through alvaar_amd/system.py: AlvaAR.set_init_pose) and by
 * __graft_entry__.smoke(). */
#ifndef ALVAAR_SYSTEM_TESTING_H
#define ALVAAR_SYSTEM_TESTING_H
#include "alvaar_system.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The two-view initialisation (VisualFrontend::initialise, src/slam/src/visual_frontend.cpp:517-549 -> MultiViewGeometry::
 * compute5ptEssentialMatrix, multi_view_geometry.cpp:225-320) adopts this pose (Twc of the initialisation frame, unit baseline) ONCE
 * instead of its own five-point result.  Why a differential test wants that: OpenGV's forward-difference refinement of the two-view pose
 * sits at its rounding-noise floor -- ONE ulp on ONE input bearing moves the reference's own result by up to 1e-4
 * (tests/test_relpose.py::test_reference_refinement_noise_floor) -- so no second build of the algorithm reproduces that pose to 1e-5, and
 * everything downstream inherits the difference as a gauge (scale / world frame) of the map.  With both maps started from the reference's
 * pose every later frame is compared at 1e-5; WITHOUT the hook the discrete state is still identical and the poses agree to 1e-3 raw /
 * 1e-4 after Sim(3) alignment (tests/test_gpu_system.py::test_system_equals_reference_without_the_hook).  NULL disarms. */
int alva_system_debug_set_init_pose(alva_system *sys, const double *pose7);

#ifdef __cplusplus
}
#endif
#endif
