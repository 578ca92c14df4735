// The nucleotide alignment kernel with a whole wavefront (64 lanes = four blocks of the reference's vectors per step) per
// alignment instead of a 16-lane group: the band of 65 cells plus block padding is then covered in two steps per
// anti-diagonal instead of five to seven.  EXPERIMENTAL: same source (nucl_core.h), checked on emulated lanes against
// the reference's vectors (tests/test_nucl_emu.py), not yet measured or parity-tested on a GPU - selected only with
// MMGPU_NUCL_LANES=64 (mmgpu_nucl_align), the 16-lane kernel is the default.
#define NUCL_NG 64
#define NUCL_NS nucl64
#define NUCL_LAUNCH launch_nucl_align64
#include "nucl_kernel.hip"
