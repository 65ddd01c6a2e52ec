/*
 * fargocpt_hip.h -- C ABI of the MI355X-native FargoCPT gas update.
 *
 * The reference (kimweiskopf/fargocpt) has no FFI layer: its per-timestep gas
 * update is a set of free C++ functions over global state, called from
 * step_Euler (src/simulation.cpp:148-267).  This header is the boundary a
 * maintainer would bind instead of those calls; every entry point names the
 * reference function(s) it replaces.  Plain C types only: pointers, sizes,
 * doubles, int32.  One context per GPU / radial slab, one host thread per
 * context, all device work on one HIP stream (fcpt_set_stream).
 *
 * Field layout is the reference's t_polargrid::Field (src/polargrid.h:111-133):
 * row-major double[nr*nphi + naz], phi contiguous.  Scalar grids have Nr rows,
 * "vector" grids (v_radial, and tau_r_phi internally) have Nr+1 rows.
 *
 * All functions return 0 on success or a negative FCPT_E* code; no exceptions
 * cross the ABI.  fcpt_last_error() gives a human-readable message.
 */
#ifndef FARGOCPT_HIP_H
#define FARGOCPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCPT_ABI_VERSION 5

/* error codes */
#define FCPT_OK 0
#define FCPT_EINVAL -1  /* bad argument / unsupported configuration */
#define FCPT_ENOMEM -2  /* host or device allocation failed */
#define FCPT_EHIP -3    /* a HIP runtime call failed */
#define FCPT_ESPLIT -4  /* slab too narrow: needs >= 2*FCPT_OVERLAP rings (src/split.cpp:42-47) */
#define FCPT_ENODEV -5  /* no HIP device */
#define FCPT_ESHEAR -6  /* a step ran beyond the FARGO shear limit without its fallback (see fcpt_step) */
#define FCPT_ECOMM -7   /* an RCCL call failed, or librccl could not be loaded */

/* src/constants.h:17-19 */
#define FCPT_OVERLAP 7
#define FCPT_GHOSTCELLS_B 1
/* rows of geometry kept past the last local ring (src/init.cpp:40-45 search_buffer) */
#define FCPT_GEOM_PAD 15
#define FCPT_MAX_BODIES 8

/* enum-like int32 values of fcpt_desc ------------------------------------ */
enum { FCPT_SPACING_ARITHMETIC = 0, FCPT_SPACING_LOGARITHMIC = 1, FCPT_SPACING_EXPONENTIAL = 2 };
enum { FCPT_EOS_ISOTHERMAL = 0, FCPT_EOS_IDEAL = 1 };
enum { FCPT_ARTVISC_NONE = 0, FCPT_ARTVISC_TW = 1, FCPT_ARTVISC_SN = 2 };
enum { FCPT_LIMITER_VANLEER = 0, FCPT_LIMITER_MC = 1 };
enum { FCPT_OPACITY_LIN = 0, FCPT_OPACITY_BELL = 1, FCPT_OPACITY_CONST = 2, FCPT_OPACITY_SIMPLE = 3 };
enum { FCPT_BETAREF_ZERO = 0, FCPT_BETAREF_REFERENCE = 1, FCPT_BETAREF_MODEL = 2, FCPT_BETAREF_FLOOR = 3 };
enum { FCPT_INTEGRATOR_EULER = 0, FCPT_INTEGRATOR_LEAPFROG = 1 };
/* per-variable boundary conditions (src/boundary_conditions/config.cpp:97-343) */
enum {
    FCPT_BC_ZEROGRADIENT = 0,
    FCPT_BC_REFERENCE = 1,
    FCPT_BC_REFLECTING = 2, /* vrad only */
    FCPT_BC_OUTFLOW = 3,    /* vrad only */
    FCPT_BC_KEPLERIAN = 4,  /* vrad: keplerian_radial, vaz: keplerian_azimuthal */
    FCPT_BC_ZEROSHEAR = 5,  /* vaz only */
    FCPT_BC_NONE = 6
};
/* damping target (src/boundary_conditions/damping.cpp:152-178) */
enum { FCPT_DAMP_NONE = 0, FCPT_DAMP_REFERENCE = 1, FCPT_DAMP_ZERO = 2, FCPT_DAMP_MEAN = 3 };
/* initial condition generator (src/init.cpp:255-343) */
enum { FCPT_IC_PROFILE = 0, FCPT_IC_SPREADING_RING = 1, FCPT_IC_SHOCKTUBE = 2 };

/* grids addressable through upload / download / device_ptr
 * (subset of t_data::t_polargrid_type, src/data.h:17-86) */
enum {
    FCPT_F_SIGMA = 0,
    FCPT_F_VRAD = 1, /* (Nr+1) x Nphi */
    FCPT_F_VAZI = 2,
    FCPT_F_ENERGY = 3,
    FCPT_F_PRESSURE = 4,
    FCPT_F_SOUNDSPEED = 5,
    FCPT_F_SCALE_HEIGHT = 6,
    FCPT_F_VISCOSITY = 7,
    FCPT_F_TEMPERATURE = 8,
    FCPT_F_POTENTIAL = 9, /* ideal EOS, Euler: evaluated inside the source step; a download fills the grid from the current state */
    FCPT_F_SIGMA0 = 10,
    FCPT_F_VRAD0 = 11, /* (Nr+1) x Nphi */
    FCPT_F_VAZI0 = 12,
    FCPT_F_ENERGY0 = 13,
    FCPT_F_QPLUS = 14,
    FCPT_F_QMINUS = 15,
    /* VISCOSITY_CORRECTION_FACTOR_PHI|R (viscosity.cpp:256-348); only with StabilizeViscosity 1|2,
     * FCPT_EINVAL otherwise */
    FCPT_F_VISC_CFAC_PHI = 16,
    FCPT_F_VISC_CFAC_R = 17,
    /* MASSFLOW (src/data.h:76), (Nr+1) x Nphi, only with fcpt_desc.write_massflow: the mass every Transport()
     * carried through the inner interface of each cell, accumulated step by step as VanLeerRadial does for the
     * density (src/TransportEuler.cpp:609-616: "+= varq_inf").  The caller divides by the elapsed
     * Nmonitor * MonitorTimestep (quantities::calculate_massflow, src/quantities.cpp:770-781), sums over azimuth
     * for MassFlow1D.dat and clears the grid by uploading zeros (clear_after_write, src/data.cpp:277). */
    FCPT_F_MASSFLOW = 18,
    /* ACCEL_RADIAL / ACCEL_AZIMUTHAL (src/data.cpp:49-58), (Nr+1) x Nphi, only with BodyForceFromPotential: no: the
     * cell-centred acceleration of the gas by the bodies and the indirect term that CalculateAccelOnGas
     * (src/Pframeforce.cpp:96-189) left for the last source step (rows 0 and Nr are never written: zero) */
    FCPT_F_ACCEL_RADIAL = 19,
    FCPT_F_ACCEL_AZIMUTHAL = 20,
    FCPT_F_COUNT = 21
};

/*
 * Everything the path reads from parameters::*, global.h and refframe:: in the
 * reference.  All lengths/times/masses are in code units (G = 1 by default).
 * The YAML key each member mirrors is given on the right
 * (src/parameters.cpp:520-900, src/Interpret.cpp:73-700,
 *  src/boundary_conditions/config.cpp, damping.cpp:185-271).
 */
typedef struct fcpt_desc {
    uint32_t struct_size; /* = sizeof(fcpt_desc), ABI check */
    uint32_t abi_version; /* = FCPT_ABI_VERSION */

    /* grid + radial slab decomposition (src/split.cpp:34-88) */
    int32_t nr_global; /* Nrad */
    int32_t nphi;      /* Naz */
    int32_t rank;      /* slab index, 0 = innermost */
    int32_t nranks;    /* number of slabs */
    int32_t radial_spacing; /* RadialSpacing */
    int32_t _pad0;
    double rmin, rmax;                   /* Rmin, Rmax */
    double exponential_cell_size_factor; /* ExponentialCellSizeFactor */

    /* equation of state */
    int32_t eos; /* EquationOfState */
    int32_t _pad1;
    double adiabatic_index; /* AdiabaticIndex */
    double mu;              /* mu */
    double aspect_ratio;    /* AspectRatio */
    double flaring_index;   /* FlaringIndex */
    double minimum_temperature, maximum_temperature; /* code units */

    /* disk profile / floors */
    double sigma0;      /* Sigma0 (code units) */
    double sigma_slope; /* SigmaSlope */
    double sigma_floor; /* SigmaFloor (multiples of sigma0) */

    /* viscosity */
    double viscous_alpha;           /* ViscousAlpha */
    double constant_viscosity;      /* ConstantViscosity */
    double radial_viscosity_factor; /* RadialViscosityFactor */
    int32_t stabilize_viscosity;    /* StabilizeViscosity: 0, 1 (damped viscous update) or 2 (dt limit) */
    int32_t artificial_viscosity;   /* ArtificialViscosity */
    double artificial_viscosity_factor;       /* ArtificialViscosityFactor */
    int32_t artificial_viscosity_dissipation; /* ArtificialViscosityDissipation */
    int32_t heating_viscous;                  /* HeatingViscous */
    double heating_viscous_factor;            /* HeatingViscousFactor */

    /* transport */
    int32_t fast_transport; /* Transport: FARGO=1 / standard=0 */
    int32_t flux_limiter;   /* FluxLimiter */

    /* time stepping */
    int32_t integrator; /* Integrator */
    int32_t _pad2;
    double cfl;                       /* CFL */
    double cfl_max_var;               /* CFLmaxVar */
    double first_dt;                  /* FirstDT */
    double heating_cooling_cfl_limit; /* HeatingCoolingCFLlimit */
    double monitor_timestep;          /* MonitorTimestep */
    int32_t nmonitor;                 /* Nmonitor */
    int32_t nsnapshots;               /* Nsnapshots */

    /* frame + gravity */
    double omega_frame;                /* OmegaFrame */
    double thickness_smoothing;        /* ThicknessSmoothing */
    int32_t body_force_from_potential; /* BodyForceFromPotential: 1 = gradient of the potential (CalculateNbodyPotential),
                                        * 0 = CalculateAccelOnGas (src/Pframeforce.cpp:96-189, SourceEuler.cpp:348-353,406-411) */
    int32_t _pad3;
    double hydro_center_mass; /* mass of the bodies defining the hydro centre */

    /* boundary conditions: [0] = inner, [1] = outer */
    int32_t bc_sigma[2];
    int32_t bc_energy[2];
    int32_t bc_vrad[2];
    int32_t bc_vaz[2];
    double keplerian_vaz_factor[2];  /* Inner/OuterBoundaryVaziKeplerianFactor */
    double keplerian_vrad_factor[2]; /* Inner/OuterBoundaryVradKeplerianFactor */

    /* wave damping */
    int32_t damping; /* Damping */
    int32_t _pad4;
    double damping_inner_limit;       /* DampingInnerLimit */
    double damping_outer_limit;       /* DampingOuterLimit */
    double damping_time_factor;       /* DampingTimeFactor */
    double damping_time_radius_outer; /* DampingTimeRadiusOuter */
    int32_t damp_vrad[2];   /* DampingVRadialInner/Outer */
    int32_t damp_vaz[2];    /* DampingVAzimuthalInner/Outer */
    int32_t damp_sigma[2];  /* DampingSurfaceDensityInner/Outer */
    int32_t damp_energy[2]; /* DampingEnergyInner/Outer */

    /* physical constants in code units (src/constants.cpp:236-262) */
    double G, Rgas, sigma_sb, c_light;

    /* initial conditions (src/init.cpp) */
    int32_t ic;                      /* ShockTube / SpreadingRing / profile */
    int32_t set_sigma0;              /* SetSigma0 */
    double disk_mass;                /* DiskMass */
    int32_t initialize_vradial_zero; /* InitializeVradialZero */
    int32_t initialize_pure_keplerian; /* InitializePureKeplerian */

    /* cooling terms of SubStep3 (calculate_qminus, src/SourceEuler.cpp:931-950; ideal EOS only) */
    int32_t cooling_surface; /* SurfaceCooling: thermal (src/SourceEuler.cpp:790-820) */
    int32_t opacity;         /* Opacity (src/opacity.cpp:10-43): FCPT_OPACITY_* (Lin :45-168, Bell :170-297, Constant, Simple) */
    double cooling_radiative_factor; /* CoolingRadiativeFactor */
    double kappa_const;              /* KappaConst (code units: L0^2 / M0) */
    double kappa_factor;             /* KappaFactor */
    double tau_factor;               /* TauFactor */
    double tau_min;                  /* TauMin */
    double density_factor;           /* DensityFactor */
    int32_t cooling_beta;            /* CoolingBetaLocal (thermal_relaxation, src/SourceEuler.cpp:632-786) */
    int32_t cooling_beta_reference;  /* CoolingBetaReference: FCPT_BETAREF_* */
    double cooling_beta_value;       /* CoolingBeta */
    double cooling_beta_ramp_up;     /* CoolingBetaRampUp (code time units) */
    /* unit system, code -> cgs (src/units.cpp:158-185), for the tabulated opacity laws */
    double temperature_cgs; /* K per code temperature unit */
    double density_cgs;     /* g/cm^3 per code volume-density unit */
    double opacity_cgs;     /* cm^2/g per code opacity unit */
    /* initial profiles: Sigma and e times 1 / (1 + exp(+-(r - point) / width)) (src/init.cpp:1063-1146,1363-1450,
     * src/util.cpp:69-93), floors re-applied */
    int32_t profile_cutoff_inner;       /* ProfileCutoffInner */
    int32_t profile_cutoff_outer;       /* ProfileCutoffOuter */
    double profile_cutoff_point_inner;  /* ProfileCutoffPointInner */
    double profile_cutoff_width_inner;  /* ProfileCutoffWidthInner */
    double profile_cutoff_point_outer;  /* ProfileCutoffPointOuter */
    double profile_cutoff_width_outer;  /* ProfileCutoffWidthOuter */
    /* monitoring that rides on the path */
    int32_t write_massflow; /* WriteMassFlow: accumulate the radial mass flux of every Transport() in FCPT_F_MASSFLOW */
    int32_t _pad5;
} fcpt_desc;

/* Row ranges of a slab, exactly the integers of src/split.cpp:56-78. */
typedef struct fcpt_split {
    int32_t nr;   /* local NRadial (overlap included) */
    int32_t imin; /* IMIN: global index of local row 0 */
    int32_t imax; /* IMAX */
    int32_t zero_no_ghost;
    int32_t one_no_ghost_vr;
    int32_t max_no_ghost;
    int32_t maxmo_no_ghost_vr;
    int32_t zero_or_active;
    int32_t max_or_active;
    int32_t radial_first_active;
    int32_t radial_active_size;
    int32_t is_first; /* CPU_Rank == 0 */
    int32_t is_last;  /* CPU_Rank == CPU_Highest */
} fcpt_split;

/* simulation clock, the state of namespace sim (src/simulation.cpp:24-31) */
typedef struct fcpt_clock {
    double time;
    double last_dt;
    uint64_t n_hydro_iter;
    uint32_t n_monitor;
    uint32_t n_snapshot;
} fcpt_clock;

typedef struct fcpt_ctx fcpt_ctx;

/* ---- host-side helpers (no GPU needed) -------------------------------- */

/* Fill a descriptor with the reference's defaults (src/parameters.cpp,
 * src/Interpret.cpp).  Always succeeds. */
int fcpt_desc_default(fcpt_desc *d);

/* SplitDomain (src/split.cpp:34-88) for slab d->rank of d->nranks. */
int fcpt_split_domain(const fcpt_desc *d, fcpt_split *out);

/* Interface radii Radii[0 .. nr_global + FCPT_GEOM_PAD] of the global grid
 * (src/init.cpp:92-145).  `radii` must hold nr_global + FCPT_GEOM_PAD + 1 doubles. */
int fcpt_radii(const fcpt_desc *d, double *radii);

/* Initial gas fields of slab d->rank (init_physics, src/init.cpp:255-343:
 * init_gas_density / init_gas_energy / init_gas_velocities / shock tube /
 * spreading ring) before the first boundary call.  With SetSigma0 the descriptor's
 * sigma0 is rescaled in place, as parameters::sigma0 is (src/init.cpp:1155).
 * Arrays are local-slab sized:
 * sigma, vazi, energy: nr*nphi; vrad: (nr+1)*nphi.  energy may be NULL for
 * isothermal runs. */
int fcpt_initial_fields(fcpt_desc *d, const double *radii, double *sigma, double *vrad,
                        double *vazi, double *energy);

const char *fcpt_last_error(void);

/* ---- context ------------------------------------------------------------ */

/* One process per GPU (the reference: one MPI rank per slab, src/parallel.cpp:28-40): the number of HIP devices this
 * process sees (0 without a GPU; never fails) and the choice of the one the following fcpt_create uses.  Call
 * fcpt_set_device before any other GPU-touching call of the process. */
int fcpt_device_count(int32_t *n);
int fcpt_set_device(int32_t device);

/* Allocate all device state for slab d->rank on the current HIP device.
 * `radii` is the global interface array from fcpt_radii (or a radii.dat).
 * Replaces data.set_size + init_radialarrays + InitTransport + cfl::init
 * (src/main.cpp:96-103, src/TransportEuler.cpp:57-96, src/cfl.cpp:14-18). */
int fcpt_create(const fcpt_desc *d, const double *radii, fcpt_ctx **out);
int fcpt_destroy(fcpt_ctx *ctx);

/* Launch all subsequent work of this context on `hip_stream` (a hipStream_t;
 * NULL = the default stream).  Lets the caller order kernels against its own
 * copies / RCCL calls without extra synchronisation. */
int fcpt_set_stream(fcpt_ctx *ctx, void *hip_stream);
int fcpt_synchronize(fcpt_ctx *ctx);

/* Kernel-selection switches of one context, by name (lower case, e.g. "transport_fallback"): which of the
 * parity-tested kernel variants the step uses, marching-chunk lengths, overlap of the ghost exchange.  The
 * environment variables FCPT_<NAME> only provide the defaults read once in fcpt_create; no launch reads the
 * environment.  -1 = the library's built-in choice.  Names: transport_fused (0 | 1 | 2), transport_rows, transport_graded, transport_big, transport_ladder, transport_rank_grade, source_graded,
 * source_rows, theta_rows, transport_fallback, transport_split, fused_source, march_source, march_source_adi,
 * theta_march, theta_fused, cfl_rings, cfl_wide_blocks, cfl_fold_in_source, gate_in_boundary, cfl_split, source_ring_parts, fused_damping, inline_potential, cfl_thermal, bc_fold, bc_in_cfl, comm_overlap,
 * comm_loopback, graph_steps, profile_stride (fcpt_profile_start times every n-th launch of the selected kernels).
 * fcpt_get_option also answers three read-only counters of fcpt_run_steps: graph_replays (hipGraphLaunch calls issued
 * so far), graph_cycle (steps per replay, 0 = no graph), coop_active (always 0: a one-kernel step was measured slower than
 * the launches it would replace), and transport_fell_back (1: the last Transport() met a ring pair
 * beyond the fused kernel's one-lane shift and took the two-kernel path; blocks).
 * FCPT_EINVAL for an unknown name.  (The reference has no counterpart: its variants are compile-time.) */
int fcpt_set_option(fcpt_ctx *ctx, const char *name, int32_t value);
int fcpt_get_option(const fcpt_ctx *ctx, const char *name, int32_t *value);

/* The chunks of rings into which the fused Transport() kernel (src/TransportEuler.cpp:112-136 as one marching pass)
 * divides the slab: one wavefront marches one tile of 53 cells over one chunk.  Where the slab needs several rounds of
 * the GPU's wavefront slots the library grades the chunk lengths (long chunks first: options transport_graded,
 * transport_big, transport_ladder), where one round covers it the lengths follow the rate at which a SIMD serves its
 * wavefronts (transport_rank_grade), so that the slots run dry together.  The chunking never changes a result -- every
 * ring of a tile is computed by exactly one wavefront from the same operands.
 * fcpt_transport_chunks reports (tile, first ring, one past the last ring) per wavefront in the order of dispatch, entries
 * with first == last being idle (n_wavefronts = 0: equal chunks of transport_rows rings); fcpt_set_transport_chunks
 * replaces the lengths by an explicit list of chunk lengths, dealt from both ends of the slab, the last entry repeating
 * (n = 0: back to the built-in choice) -- a tuning and test hook, like the options.
 * (No counterpart in the reference: its loops are not chunked.) */
int fcpt_set_transport_chunks(fcpt_ctx *ctx, const int32_t *lengths, int32_t n);
int fcpt_transport_chunks(const fcpt_ctx *ctx, int32_t *tile_first_last, int32_t capacity, int32_t *n_wavefronts);
/* Likewise the marching kernels of the source step (update_with_sourceterms ... SubStep3 as one pass, src/SourceEuler.cpp,
 * src/viscosity/): where one round of wavefronts covers the slab, every wavefront's chunk length is matched to the rate at
 * which its SIMD will serve it (option source_graded, per cent of difference between first and last rank; 0: equal
 * chunks).  Reports (segment of 59 cells, first ring, one past the last ring) per wavefront in the order of dispatch,
 * entries with first == last being idle; n_wavefronts = 0: equal chunks of source_rows rings. */
int fcpt_source_chunks(const fcpt_ctx *ctx, int32_t *seg_first_last, int32_t capacity, int32_t *n_wavefronts);

int fcpt_get_split(const fcpt_ctx *ctx, fcpt_split *out);
int fcpt_get_clock(const fcpt_ctx *ctx, fcpt_clock *out);
int fcpt_set_clock(fcpt_ctx *ctx, const fcpt_clock *in);
/* hydro_dt_logger (src/hydro_dt_logger.h:13-34: "min dt" / "max dt" of monitor/timestepLogging.dat): the smallest and
 * largest step length since the last call with reset != 0, kept next to the device clock so that callers who let
 * fcpt_run_steps run many steps without reading dt still get them.  Blocks. */
int fcpt_dt_statistics(fcpt_ctx *ctx, double *dt_min, double *dt_max, int32_t reset);

/* Host <-> device copies of one grid in the reference's Field layout
 * (what read2D / write2D move, src/polargrid.cpp:135-180,301-353).
 * Synchronous with respect to the context's stream. */
int fcpt_upload(fcpt_ctx *ctx, int32_t field, const double *host);
int fcpt_download(fcpt_ctx *ctx, int32_t field, double *host);
/* Device address and element count of a grid, for zero-copy callers.  The transport is out of place:
 * the addresses of SIGMA, ENERGY, VRAD and VAZI may change in fcpt_step / fcpt_step_device /
 * fcpt_run_steps -- query again after a step. */
int fcpt_device_ptr(fcpt_ctx *ctx, int32_t field, void **dptr, uint64_t *count);

/* Positions, masses and cubic smoothing radii of the N-body objects for the
 * next potential evaluation, plus the indirect-term acceleration
 * (CalculateNbodyPotential, src/Pframeforce.cpp:21-94).  Body 0 is the star. */
int fcpt_set_bodies(fcpt_ctx *ctx, int32_t n, const double *x, const double *y, const double *mass,
                    const double *cubic_smoothing_radius, double indirect_x, double indirect_y);

/* Leapfrog only: the bodies at the mid-step time, used for the potential of the second gas
 * kick (src/simulation.cpp:359-366).  Without this call the second kick reuses the positions of
 * fcpt_set_bodies.  Must follow fcpt_set_bodies (same n). */
int fcpt_set_bodies_midstep(fcpt_ctx *ctx, int32_t n, const double *x, const double *y, const double *mass,
                            const double *cubic_smoothing_radius);

/* Stellar irradiation of the disk in SubStep3 (irradiation_single, src/SourceEuler.cpp:538-612): body k
 * of fcpt_set_bodies heats the gas if temperature[k] > 0 (code units), with its radius (code units)
 * and the ramp-up time of the heating.  Any irradiating body also switches the effective optical
 * depth to the irradiated form (kappa_eff, src/compute.cpp:64-77).  Ideal EOS only. */
int fcpt_set_body_irradiation(fcpt_ctx *ctx, int32_t n, const double *temperature, const double *radius,
                              const double *rampup_time);

/* Specific force of this slab's gas on an object at (x, y): ComputeDiskOnPlanetAccel
 * (src/Force.cpp:23-122) without its MPI_Allreduce -- out = {a_x, a_y} summed over the cells
 * of the rings inside `r_object` (the object's distance to the origin) followed by {a_x, a_y}
 * of the rings outside, over this slab's active rings; the caller adds the slabs' four sums
 * (the reference's 4-double all-reduce) and then inner + outer.  Smoothing as
 * compute_smoothing (src/Force.cpp:124-159): `smoothing_fixed` < 0 selects
 * ThicknessSmoothing x H of each cell, a value >= 0 is used as it is (planet-location smoothing,
 * or 0 for the star with compatibility_no_star_smoothing).  `cubic_smoothing_radius` > 0 applies
 * the derivative of the Klahr & Kley cubic smoothing inside that distance.
 * Blocks until the four sums are on the host. */
int fcpt_disk_on_body_accel(fcpt_ctx *ctx, double x, double y, double r_object, double smoothing_fixed,
                            double cubic_smoothing_radius, double out[4]);

/* After the initial Sigma/v/energy upload: init_euler (src/SourceEuler.cpp:251-285:
 * sound speed, pressure, temperature, scale height, viscosity), the first
 * potential, copy_initial_values + apply_boundary_condition + copy_initial_values
 * (src/init.cpp:337-341). */
int fcpt_init_physics(fcpt_ctx *ctx);

/* recalculate_derived_disk_quantities (src/SourceEuler.cpp:225-249) after the state grids were replaced from
 * outside (restart_load, src/restart.cpp:18-139: uploads of Sigma, vrad, vazi, energy, Qplus, Qminus and of the
 * t = 0 grids into FCPT_F_*0, then this call). */
int fcpt_recalculate_derived(fcpt_ctx *ctx);

/* ---- the hot path -------------------------------------------------------- */

/* cfl::condition_cfl without the MPI_Allreduce (src/cfl.cpp:185-376): the
 * minimum over this slab's active rings.  Blocks until the value is on the host. */
int fcpt_cfl(fcpt_ctx *ctx, double *dt_local);

/* Device-resident variants for multi-slab runs that keep dt off the host: fcpt_cfl_device
 * writes this slab's CFL minimum to *d_dt_local (a device address, e.g. of a tensor the
 * caller then MIN-all-reduces with RCCL on the same stream); fcpt_calculate_timestep_device
 * applies the CalculateTimeStep policy to the reduced value read from *d_cfl_global (NULL: the value
 * fcpt_cfl_allreduce left in the library's own device scalar) and
 * leaves the step length in the device clock, where fcpt_step_device / fcpt_post_device
 * pick it up.  No call in this group synchronises with the host. */
int fcpt_cfl_device(fcpt_ctx *ctx, double *d_dt_local);
int fcpt_calculate_timestep_device(fcpt_ctx *ctx, const double *d_cfl_global);
int fcpt_step_device(fcpt_ctx *ctx);
int fcpt_post_device(fcpt_ctx *ctx);

/* fcpt_step_device for slabs with neighbours, in two halves around the caller's ghost exchange: _begin queues
 * the source step and marches the chunks of the transport that hold the rings the neighbours are waiting for
 * (rows [7,14) and [nr-14,nr-7)) on the context's stream and all other chunks on an internal stream; the
 * caller then queues fcpt_exchange_pack and its transfers, which run under the interior chunks; _end makes the
 * context's stream wait for them.  Between the two only fcpt_exchange_pack may be called.  Same results as
 * fcpt_step_device (the chunks are independent); falls back to it (and _end is a no-op) when the one-kernel
 * transport does not apply: leapfrog, rings shorter than 256 cells, a dt that is not the CFL policy's. */
int fcpt_step_device_begin(fcpt_ctx *ctx);
int fcpt_step_device_end(fcpt_ctx *ctx);

/* Hides the ghost exchange behind the next step's CFL reduction (slabs with neighbours): queued after
 * fcpt_step* and fcpt_exchange_pack, before the caller waits for the neighbours' rings, it evaluates
 * condition_cfl (src/cfl.cpp:222-330) on the rings that neither fcpt_exchange_unpack nor the boundary kernels
 * write (rows [8, nr-9)); the next fcpt_cfl / fcpt_cfl_device then only adds the rings next to the ghost zones
 * and reduces.  Same dt as the unsplit call.  A no-op whenever the interior is not yet final at that point
 * (damping outside the step kernels, fields uploaded since) -- the later call then does all rings. */
int fcpt_cfl_begin(fcpt_ctx *ctx);

/* sim::CalculateTimeStep's policy (src/simulation.cpp:100-118) applied to the
 * globally reduced CFL dt: returns min(CFLmaxVar*last_dt, cfl_dt) and stores it
 * as last_dt. */
int fcpt_calculate_timestep(fcpt_ctx *ctx, double cfl_dt_global, double *dt);

/* The monitor-time snapping of sim::run (src/simulation.cpp:528-540). */
int fcpt_snap_to_monitor(const fcpt_ctx *ctx, double cfl_dt, double *step_dt);

/* The gas part of one step.  Integrator: Euler -- step_Euler up to and including Transport (src/simulation.cpp:167-217):
 * potential, update_with_sourceterms, update_with_artificial_viscosity,
 * recalculate_viscosity, compute_viscous_stress_tensor,
 * update_velocities_with_viscosity, SubStep3, apply_boundary_condition(final=false),
 * Transport.  Integrator: Leapfrog -- step_LeapFrog (src/simulation.cpp:316-393): gas kick 1/2 with
 * dt/2, boundary (final=false), Transport with dt, potential at mid-step, compute_pressure,
 * gas kick 2/2 with dt/2.  Asynchronous on the context's stream.  Advances time by dt.
 *
 * The fused transport kernel handles |Nshift[i] - Nshift[i-1]| <= 1, which the FARGO shear term of the CFL
 * condition (src/cfl.cpp:207-220) makes the normal case; the two-kernel transport is queued behind it as a
 * device-side fallback and runs only when a ring pair exceeds that (a dt beyond the CFL step, or a source step
 * that changed v_phi violently), so every step is computed as the reference does.  FCPT_TRANSPORT_FALLBACK=0
 * (environment) drops the two idle launches for flows known to be benign; if the limit is then ever exceeded,
 * the next synchronising call returns FCPT_ESHEAR. */
int fcpt_step(fcpt_ctx *ctx, double dt);

/* CommunicateBoundaries, device side (src/commbound.cpp:108-125,163-180):
 * pack rows [7,14) -> send_inner and rows [nr-14,nr-7) -> send_outer of
 * Sigma, vrad, vazi(, energy); unpack recv_inner -> rows [0,7) and
 * recv_outer -> rows [nr-7,nr).  Buffers hold fcpt_exchange_count() doubles
 * each; device buffers take one copy kernel per call, host buffers one
 * hipMemcpyAsync per field and side; a NULL pointer skips that side.  The
 * transfer itself (RCCL send/recv between neighbouring slabs) is done by the
 * caller on the same stream. */
int fcpt_exchange_count(const fcpt_ctx *ctx, uint64_t *count);
int fcpt_exchange_pack(fcpt_ctx *ctx, double *send_inner, double *send_outer);
int fcpt_exchange_unpack(fcpt_ctx *ctx, const double *recv_inner, const double *recv_outer);

/* ---- radial slabs on several GPUs: RCCL inside the library ------------------------------------------
 * The reference's two communication points, on the context's stream (no host synchronisation, no stream hop):
 * CommunicateBoundaries' MPI_Isend/MPI_Irecv with CPU_Prev/CPU_Next (src/commbound.cpp:130-158) become one
 * group of ncclSend/ncclRecv with slab rank-1 / rank+1 over xGMI, condition_cfl's MPI_Allreduce(MPI_MIN)
 * (src/cfl.cpp:379) an ncclAllReduce(ncclMin) of one device double.  One process (or thread) per GPU, as one
 * MPI rank per slab in the reference (src/parallel.cpp:28-40).
 *
 * Bootstrap = ncclGetUniqueId on slab 0, the 128 bytes handed to every slab by the host's own means (the
 * reference's host would MPI_Bcast them; bench.py uses its torch.distributed store, the C++ driver a file),
 * then fcpt_comm_init on every slab (collective: ncclCommInitRank with rank = d->rank of d->nranks, on the
 * device the context was created on).  librccl is bound at run time; single-GPU users never load it. */
#define FCPT_COMM_ID_BYTES 128
int fcpt_comm_unique_id(void *id128);
int fcpt_comm_init(fcpt_ctx *ctx, const void *id128);
int fcpt_comm_destroy(fcpt_ctx *ctx);

/* The same communicator for ranks that cannot use RCCL among themselves -- several slabs on one GPU (RCCL refuses two
 * ranks on one device), i.e. rehearsals and tests of the N-rank path on a 1-GPU box, or more slabs than devices:
 * ghost rings and the CFL value are staged through pinned host memory and the shared-memory file `path` (collective
 * over the d->nranks slabs of ONE node; slab 0 creates the file, which must not belong to another run, and removes it
 * in fcpt_comm_destroy).  fcpt_exchange, fcpt_cfl_allreduce, fcpt_comm_barrier and fcpt_run_steps then work as with
 * fcpt_comm_init, but block the host in every transfer.  FCPT_HOSTLINK_TIMEOUT (seconds, default 120) bounds every
 * wait for another rank: FCPT_ECOMM afterwards. */
int fcpt_comm_init_host(fcpt_ctx *ctx, const char *path);

/* MPI_Barrier over the slabs of the communicator (src/main.cpp:141, the output routines): returns when every slab
 * has called it; the work queued on this context's stream before the call is complete then. */
int fcpt_comm_barrier(fcpt_ctx *ctx);

/* CommunicateBoundaries (src/commbound.cpp:98-182) of this slab, whole: pack rows [7,14) / [nr-14,nr-7),
 * grouped send/recv with the neighbours, unpack into rows [0,7) / [nr-7,nr).  Asynchronous.  With the option
 * comm_overlap the transfers run on the library's communication stream while the CFL terms of the interior
 * rings (fcpt_cfl_begin) are evaluated on the context's stream. */
int fcpt_exchange(fcpt_ctx *ctx);

/* condition_cfl's MPI_Allreduce (src/cfl.cpp:379), device-resident: queues the local reduction
 * (src/cfl.cpp:185-376) and the MIN over all slabs; the result stays in device memory, where
 * fcpt_calculate_timestep_device(ctx, NULL) picks it up.  With dt_global != NULL the call blocks and also
 * returns the value.  fcpt_run_steps uses both calls when the context has a communicator. */
int fcpt_cfl_allreduce(fcpt_ctx *ctx, double *dt_global);

/* The rest of step_Euler after CommunicateBoundaries (src/simulation.cpp:244-265):
 * apply_boundary_condition(final=true) (damping first) and
 * recalculate_derived_disk_quantities.  Asynchronous. */
int fcpt_post(fcpt_ctx *ctx, double dt);

/* boundary_conditions::apply_boundary_condition (src/boundary_conditions/boundary_conditions.cpp:65-114):
 * wave damping first when `final` is non-zero, then the eight per-variable ghost-ring
 * conditions.  Asynchronous.  (fcpt_step and fcpt_post call it internally; sim::init
 * calls it once more before the loop, src/simulation.cpp:463.) */
int fcpt_apply_boundary(fcpt_ctx *ctx, double dt, int32_t final);

/* K iterations of {cfl [+ MIN over the slabs], calculate_timestep, [snap], step, [exchange], post} without
 * returning to the caller, as sim::run does (src/simulation.cpp:515-553).  A context with a communicator
 * (fcpt_comm_init) runs the multi-slab loop: every slab's host calls this with the same arguments.  If `snap` is non-zero the
 * step is snapped to monitor times and n_monitor advances.  nsteps_done may be
 * NULL.  Stops early when time reaches t_final = nsnapshots*nmonitor*monitor_timestep
 * unless t_final <= 0. */
int fcpt_run_steps(fcpt_ctx *ctx, int64_t nsteps, int32_t snap, int64_t *nsteps_done);

/* ---- measurement ----------------------------------------------------------- */

/* Per-kernel timing with HIP events recorded on the context's stream around each
 * launch of the kernels selected by `mask` (bit k = kernel id k, see
 * fcpt_kernel_name).  At most max_launches launches are recorded between start and
 * stop.  fcpt_profile_stop synchronises and fills ms_total[id] / launches[id]
 * (arrays of fcpt_kernel_count() entries).  Replaces nothing in the reference (its
 * only timer is the wall clock of hydro_dt_logger, src/hydro_dt_logger.h:13-34). */
int32_t fcpt_kernel_count(void);
const char *fcpt_kernel_name(int32_t id);
int fcpt_profile_start(fcpt_ctx *ctx, uint64_t mask, int32_t max_launches);
int fcpt_profile_stop(fcpt_ctx *ctx, double *ms_total, int64_t *launches);

/* Test hook, needs no GPU: the tables fcpt_transport_chunks / fcpt_source_chunks would report for a slab of nr x nphi
 * cells on a device of n_cu compute units (8 XCDs), isothermal or ideal EOS, with damping zones of damp_inner / damp_outer
 * rings folded into the transport.  n_transport / n_source = 0: equal chunks (small grids, rings of > 8192 cells ...). */
int fcpt_selftest_chunk_tables(int32_t nr, int32_t nphi, int32_t n_cu, int32_t adiabatic, int32_t damp_inner, int32_t damp_outer,
                               int32_t *transport_tile_first_last, int32_t transport_capacity, int32_t *n_transport,
                               int32_t *source_seg_first_last, int32_t source_capacity, int32_t *n_source);
/* Test hook: out[k] = 0.5 * flux_limiter(a[k], b[k]) (src/TransportEuler.cpp:306-337; limiter = FCPT_LIMITER_*) exactly
 * as the transport kernels evaluate it on the device -- the van Leer form there is branch-free (max(ab, 0) times a
 * guarded reciprocal of a + b) and its edge cases (+-0, denormal and cancelling sums) are what this call lets a test
 * compare with `ab > 0 ? ab / (a + b) : 0`.  Host arrays of n doubles; blocks. */
int fcpt_selftest_half_limiter(int32_t limiter, int64_t n, const double *a, const double *b, double *out);

#ifdef __cplusplus
}
#endif
#endif /* FARGOCPT_HIP_H */
