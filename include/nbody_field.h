/* nbody_field.h — C-ABI of the snapshot -> density-field chain in libnbody_amd.so.
 *
 * What it replaces in the reference (pure-Python / numba loops + numpy FFTs):
 *   cic_density(positions, boxsize, N)       TheFeedingPipeline/FraP_kTilDelta.py:93-123, SnapVsPK/Power_compare.py:100-132
 *   compute_delta(density)                   TheFeedingPipeline/FraP_kTilDelta.py:126-127
 *   deconvolve_cic(delta_grid, boxsize, N)   TheFeedingPipeline/FraP_kTilDelta.py:73-91, SnapVsPK/Power_compare.py:80-97
 *   compute_power_spectrum_from_snapshot     SnapVsPK/Power_compare.py:135-200
 * driven per snapshot by TheFeedingPipeline/Feeder_pipe.py:163-195.
 *
 * A plan owns an ngrid^3 real grid and an ngrid^3 complex grid in HBM plus a hipFFT Z2Z plan. The small 1-D tables
 * (CIC window along one axis, k values along one axis, histogram edges) are computed by the HOST exactly as the
 * reference computes them (numpy) and handed over, so that bin assignment is bit-identical; all O(N_particles) and
 * O(ngrid^3) work runs on the GPU. Every function returns 0 on success, <0 (NB_ERR_* of nbody_amd.h) on failure;
 * nbf_last_error gives the message. Host pointers unless stated otherwise; a plan is not thread-safe.
 */
#ifndef NBODY_FIELD_H
#define NBODY_FIELD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbf_plan nbf_plan;

int nbf_create(nbf_plan** out, int ngrid, double boxsize, int device);
int nbf_destroy(nbf_plan* plan);
const char* nbf_last_error(const nbf_plan* plan);

/* w1d[ngrid] = sinc(k/(2 k_Nyq)) along one axis (FraP_kTilDelta.py:73-81); W = ((w[i]*w[j])*w[k])^2, zeros -> 1 */
int nbf_set_window(nbf_plan* plan, const double* w1d);
/* kvals[ngrid] = 2*pi*fftfreq(ngrid, d); edges[nedges] = histogram bin edges (Power_compare.py:179-191) */
int nbf_set_kbins(nbf_plan* plan, const double* kvals, const double* edges, int nedges);

/* grid <- CIC deposit of n particles, pos = (n,3) f64 array-of-vectors as in `components/matter/pos` */
int nbf_cic(nbf_plan* plan, const double* pos, int64_t n);
/* mean of the real grid (fixed-order fp64 reduction) */
int nbf_grid_mean(nbf_plan* plan, double* mean);
/* grid <- (grid / div1) / div2 - sub      (compute_delta: div1 = mean, div2 = 1, sub = 1) */
int nbf_grid_affine(nbf_plan* plan, double div1, double div2, double sub);
/* grid <- real(ifftn(fftn(grid) / W)) */
int nbf_deconvolve(nbf_plan* plan);
/* the same with the window clipped from below, W = max(W, clip) — the estimator variant of 1LPTvs2LPT/1vs2tlptplotter.py:81-116 and
 * Different_Sizes/Plotter.py:148-202 (`np.clip(W_CIC, 1e-3, None)`); clip = 0 is nbf_deconvolve */
int nbf_deconvolve_clip(nbf_plan* plan, double clip);
/* grid <- (double)(float)grid: the density of the scripts that deposit into a float32 array (Different_Sizes/Plotter.py:167) */
int nbf_grid_round_f32(nbf_plan* plan);
/* copy the real grid out / in (ngrid^3 doubles, C order [i][j][k]) */
int nbf_get_grid(nbf_plan* plan, double* out);
int nbf_set_grid(nbf_plan* plan, const double* in);
/* psum[nedges-1], counts[nedges-1] <- histogram over all modes of |fftn(grid)|^2 * scale, binned by |k| */
int nbf_power_bins(nbf_plan* plan, double scale, double* psum, int64_t* counts);

/* |delta_k / W|^2 L^3 / N^6 for every mode of the CIC-deposited overdensity, in np.fft order (out: host, ngrid^3 doubles) — CONCEPT's
 * estimator settings 'interpolation': 'CIC', 'deconvolve': True and, with interlace != 0, 'interlace': True (NN.param:30-41: a second
 * lattice displaced by half a cell along every axis, combined in k-space — cancels the odd aliases). Both lattices go through ONE
 * complex FFT. Replaces the host FFTs of the estimator behind the CONCEPT-style power-spectrum tables (PPPMP3M/PP/powerspec/powerspec_a=1.00). */
int nbf_power_modes(nbf_plan* plan, const double* pos, int64_t n, int interlace, double* out);

/* grad(phi2) of second-order Lagrangian perturbation theory from the first-order displacement psi = -grad(phi1) (psi, out: host, [3][ngrid^3]):
 * laplace(phi2) = sum_{i<j} [ phi1,ii phi1,jj - (phi1,ij)^2 ], all derivatives spectral — CONCEPT's `realization_options = {'LPT': 2}`
 * (Different_Sizes/Giant.param:6,70-72; 1LPTvs2LPT/1vs2tlptplotter.py:121-123 compares the two orders). 13 hipFFT transforms. */
int nbf_lpt2_gradient(nbf_plan* plan, const double* psi, double* out);

/* ---- Gaussian-random-field + Zel'dovich initial conditions --------------------------------
 * Replaces the O(N^3) Python triple loop createR (TheFeedingPipeline/Cosmo data create (new).py:24-47,
 * CiC/3D startbetingelser1.py:82-94) and the field synthesis around it (Cosmo data create (new).py:129-135,
 * CiC/3D startbetingelser1.py:98-112, 119-173). The Gaussian draws c1, c2 and the amplitude grid sqrt(P(k))*fac are
 * host inputs (the reference's RNG and CLASS stay on the host); everything O(ngrid^3) runs on the GPU. */
/* R <- Hermitian-symmetrised c1 + i*c2 (host arrays, ngrid^3 each). variant 0: self-conjugate points are replaced
 * by their own conjugate (Cosmo data create (new).py:39-45); variant 1: they are left alone (3D startbetingelser1.py:86-94) */
int nbf_noise_hermitian(nbf_plan* plan, const double* c1, const double* c2, int variant);
/* copy R out as interleaved (re, im) doubles, 2*ngrid^3 */
int nbf_get_noise(nbf_plan* plan, double* out_re_im);
/* amplitude grid delta_k = sqrt(P_k_grid) * fac (host, ngrid^3) and the k values along one axis (host, ngrid) */
int nbf_set_amplitude(nbf_plan* plan, const double* amp, const double* kvals);
/* grid <- real(ifftn(amp * mult * R)); component 0: delta (mult = 1); 1/2/3: psi_x/y/z (mult = i k_c / |k|^2,
 * |k| at [0,0,0] replaced by 1e-10 as in the reference) */
int nbf_grf_field(nbf_plan* plan, int component);
/* pos_out[(ngrid^3),3] <- (lattice + psi) mod L, lattice = i*L/ngrid on every axis (host output) */
int nbf_zeldovich(nbf_plan* plan, double* pos_out);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_FIELD_H */
