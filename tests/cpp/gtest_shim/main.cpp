// main() of the drop-in proof: the reference's own TEST_*.cpp, compiled unchanged against include/compat, run here
// through libsdrx on the GPU.  Usage: dropin_ref_tests [substring filter].  Exit status = any test failed.
#define SDRX_GTEST_MAIN
#include <gtest/gtest.h>
