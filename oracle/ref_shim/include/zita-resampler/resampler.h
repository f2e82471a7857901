/* oracle/ref_shim: zita-resampler is not available in this image.  The stand-in is the oracle's restatement of the
 * library's published algorithm (oracle/zita_restated.h; PARITY UNPINNED against the real library), so that the
 * unmodified reference sources can run their resampling paths (resample.cc, wavchunkloader.cc, wmspeed.cc).
 * TEST INFRASTRUCTURE ONLY. */
#pragma once
#include "../../../zita_restated.h"
class Resampler : public ZitaResampler
{
};
