// placeholder until the lanes-as-amplitudes family lands
#include "qc_internal.h"
int qc_wave_value_fwd(const qc_program*, const QcTrig*, const float*, const float*, float*, int64_t, hipStream_t) { return QC_ERR_UNSUPPORTED; }
int qc_wave_value_bwd(const qc_program*, const QcTrig*, const float*, const float*, const float*, float*, float*, int64_t, int64_t, int64_t, hipStream_t) { return QC_ERR_UNSUPPORTED; }
int qc_wave_jets_fwd(const qc_program*, const QcTrig*, const float*, const float*, float*, int64_t, hipStream_t) { return QC_ERR_UNSUPPORTED; }
int qc_wave_jets_bwd(const qc_program*, const QcTrig*, const float*, const float*, const float*, float*, float*, int64_t, int64_t, int64_t, hipStream_t) { return QC_ERR_UNSUPPORTED; }
