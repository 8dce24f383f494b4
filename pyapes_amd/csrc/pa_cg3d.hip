// pa_cg3d.hip -- 3-D fast path of the two CG phases (placeholder: not covered yet)
#include "pa_host.h"

template <typename T>
int pa_cg3d_phase_a(pa_ctx*, const DevEq<T>&, Vec<T>, Vec<T>, T*, double*) { return 0; }
template <typename T>
int pa_cg3d_phase_b(pa_ctx*, const DevEq<T>&, Vec<T>, T*, T*, double*) { return 0; }

template int pa_cg3d_phase_a<float>(pa_ctx*, const DevEq<float>&, Vec<float>, Vec<float>, float*, double*);
template int pa_cg3d_phase_a<double>(pa_ctx*, const DevEq<double>&, Vec<double>, Vec<double>, double*, double*);
template int pa_cg3d_phase_b<float>(pa_ctx*, const DevEq<float>&, Vec<float>, float*, float*, double*);
template int pa_cg3d_phase_b<double>(pa_ctx*, const DevEq<double>&, Vec<double>, double*, double*, double*);
