// placeholder until the MFMA kernel lands
#include "common.hpp"
namespace pds {
bool conv2d_mfma_supported(const ConvLayer&) { return false; }
int conv2d_mfma_tiles(const Geom&) { return 0; }
int launch_conv2d_mfma(const ConvLayer&, hipStream_t) { return set_error(-1, "conv2d_mfma: not built"); }
}
