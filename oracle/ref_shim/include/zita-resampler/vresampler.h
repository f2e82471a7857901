/* oracle/ref_shim: see resampler.h */
#pragma once
#include "../../../zita_restated.h"
class VResampler : public ZitaVResampler
{
};
