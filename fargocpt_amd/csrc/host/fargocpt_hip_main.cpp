// fargocpt_hip -- host driver over the C ABI: reads a FargoCPT YAML setup, runs the gas update
// on the GPU with the reference's main loop and writes snapshots in the reference's format.
//
//   fargocpt_hip [-q] [--lenient] [-N <steps>] [--ranks <n>] [--transport auto|rccl|host] start|auto|restart [N] <config.yml>
//
// --ranks n: the reference's `mpirun -np n fargocpt_exe ...` (src/parallel.cpp:28-40, one MPI rank per radial slab):
// this process starts n copies of itself -- before anything touches a GPU -- one per slab and GPU, and waits for
// them; the slabs exchange their ghost rings and the CFL minimum through the library (RCCL when every rank has a GPU
// of its own, the host-staged transport otherwise, e.g. 2 or 3 ranks on a 1-GPU box) and write their windows of the
// snapshot files as write2D does (src/polargrid.cpp:135-180).  A launcher of one's own sets FCPT_RANK, FCPT_NRANKS
// and FCPT_RDV (a directory all ranks see) instead.
//
// Counterpart of src/main.cpp:47-164 + sim::run (src/simulation.cpp:505-558) +
// handle_outputs (:50-98) for the gas path only.  N-body objects do not feel anything here
// (no REBOUND, no disk feedback): the star sits at the origin and every planet moves on its
// initial circular orbit, seen in the frame rotating with OmegaFrame.
//
// Supported configuration keys: the ones the hot path consumes (SURVEY.md section 5 "Config /
// flags"); unit suffixes au / solMass / jupiterMass / earthMass / g/cm2 / K are understood
// for the default unit system (l0 = 1 au, m0 = 1 solMass).  A key the reference's reader does not know is
// fatal, as in the reference (src/config.cpp:134-138: a typo must not silently change the physics); --lenient
// downgrades that to a warning.  Keys the reference knows but the gas path does not consume are accepted (listed
// unless -q); features outside the path that a setup switches on (self-gravity, particles, FLD ...) are refused.
#include "../../../include/fargocpt_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

namespace {

std::string lower(std::string s)
{
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}
std::string trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos)
        return "";
    const size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}
std::string unquote(std::string v)
{
    v = trim(v);
    if (v.size() >= 2 && (v.front() == '\'' || v.front() == '"') && v.back() == v.front())
        v = v.substr(1, v.size() - 2);
    return v;
}

// "key: value  # comment" lines and one list of maps ("nbody:"), which is all the reference's
// setups use (src/config.cpp wraps yaml-cpp; keys are case-insensitive, :344-349).
struct Config {
    std::map<std::string, std::string> kv;
    std::vector<std::map<std::string, std::string>> nbody;
    mutable std::map<std::string, bool> used;

    bool load(const std::string &path)
    {
        std::ifstream f(path);
        if (!f)
            return false;
        std::string line;
        bool in_nbody = false;
        while (std::getline(f, line)) {
            // strip comments (a '#' preceded by whitespace or at column 0)
            for (size_t i = 0; i < line.size(); ++i)
                if (line[i] == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) {
                    line = line.substr(0, i);
                    break;
                }
            if (trim(line).empty())
                continue;
            const bool indented = line[0] == ' ' || line[0] == '\t' || line[0] == '-';
            std::string t = trim(line);
            if (in_nbody && indented) {
                if (t[0] == '-') {
                    nbody.emplace_back();
                    t = trim(t.substr(1));
                }
                const size_t c = t.find(':');
                if (c != std::string::npos && !nbody.empty())
                    nbody.back()[lower(trim(t.substr(0, c)))] = unquote(t.substr(c + 1));
                continue;
            }
            in_nbody = false;
            const size_t c = t.find(':');
            if (c == std::string::npos)
                continue;
            const std::string key = lower(trim(t.substr(0, c)));
            const std::string val = unquote(t.substr(c + 1));
            if (key == "nbody") {
                in_nbody = true;
                continue;
            }
            kv[key] = val;
        }
        return true;
    }
    bool has(const std::string &k) const { return kv.count(lower(k)) > 0; }
    std::string str(const std::string &k, const std::string &def) const
    {
        used[lower(k)] = true;
        auto it = kv.find(lower(k));
        return it == kv.end() ? def : it->second;
    }
    bool flag(const std::string &k, bool def) const
    {
        const std::string v = lower(str(k, def ? "yes" : "no"));
        return !v.empty() && (v[0] == 'y' || v[0] == 't' || v[0] == '1');
    }
};

// Every top-level key the reference's own reader visits (config::cfg.get* / contains calls of src/parameters.cpp,
// src/Interpret.cpp, src/units.cpp, src/boundary_conditions/{config,damping}.cpp, src/particles, src/fld ..., lower
// case): a key outside this set is what src/config.cpp:134-138 dies on ("Unknown key(s) found in config file").
const char *const kReferenceKeys[] = {
    "accretewithoutdiskfeedback", "adiabatic", "adiabaticindex", "alphacold", "alphahot", "alphamode",
    "artificialviscosity", "artificialviscositydissipation", "artificialviscosityfactor", "aspectratio",
    "aspectratiomode", "bitwiseexactrestarting", "bodyforcefrompotential", "cartesianparticles",
    "centerprofiledensitycorrectionfactor", "cfl", "cflmaxvar", "cicplanet", "circumbinarydecayexponent",
    "circumbinarydecaywidth", "circumbinaryring", "circumbinaryringenhancementfactor", "circumbinaryringposition",
    "circumbinaryringwidth", "compatibilitynostarsmoothing", "compatibilitysmoothingplanetloc", "constantviscosity",
    "coolingbeta", "coolingbetalocal", "coolingbetarampup", "coolingbetareference", "coolingbetaziampras2023",
    "coolingbetaziampras2023method", "coolingradiativefactor", "corotationreferencebody", "correctdiskselfgravity",
    "cps", "cvnr", "damping", "dampingenergyinner", "dampingenergyouter", "dampinginnerlimit", "dampingouterlimit",
    "dampingsurfacedensityinner", "dampingsurfacedensityouter", "dampingtimefactor", "dampingtimeradiusouter",
    "dampingvazimuthalinner", "dampingvazimuthalouter", "dampingvradialinner", "dampingvradialouter",
    "densityfactor", "disk", "diskfeedback", "diskmass", "diskradiusmassfraction", "dowrite1dfiles",
    "energycondition", "energyfilename", "equationofstate", "exponentialcellsizefactor", "featuresize", "firstdt",
    "flaringindex", "fluxlimiter", "frame", "heatingcoolingcfllimit", "heatingviscous", "heatingviscousfactor",
    "hydroframecenter", "hydrogenmassfraction", "imposeddiskdrift", "indirecttermdiskondisk", "indirecttermmode",
    "initializepurekeplerian", "initializevradialzero", "innerboundary", "innerboundaryenergy", "innerboundarysigma",
    "innerboundaryvazi", "innerboundaryvazikeplerianfactor", "innerboundaryvrad", "innerboundaryvradkeplerianfactor",
    "integrateparticles", "integrator", "kappaconst", "kappafactor", "keepdiskmassconstant", "klahrsmoothingradius",
    "l0", "logafterrealseconds", "logaftersteps", "m0", "massaccretionradius", "maximumtemperature",
    "minimumtemperature", "monitortimestep", "mu", "naz", "nbody", "nmonitor", "nrad", "nsnapshots",
    "numberofparticles", "omegaframe", "opacity", "outerboundary", "outerboundaryenergy", "outerboundarysigma",
    "outerboundaryvazi", "outerboundaryvazikeplerianfactor", "outerboundaryvrad", "outerboundaryvradkeplerianfactor",
    "outputdir", "particledensity", "particlediskgravityenabled", "particledustdiffusion", "particleeccentricity",
    "particlegasdragenabled", "particleintegrator", "particlemaximumescaperadius", "particlemaximumradius",
    "particleminimumescaperadius", "particleminimumradius", "particleradius", "particleradiusincreasefactor",
    "particlespeciesnumber", "particlesurfacedensityslope", "planetorbitdisktest", "polytropicconstant",
    "profilecutoffinner", "profilecutoffouter", "profilecutoffpointinner", "profilecutoffpointouter",
    "profilecutoffwidthinner", "profilecutoffwidthouter", "quantitiesradiuslimit", "radialspacing",
    "radialviscosityfactor", "radiativediffusion", "radiativediffusionautoomega", "radiativediffusionchecksolution",
    "radiativediffusiondumpdata", "radiativediffusioninnerboundary", "radiativediffusionmaxiterations",
    "radiativediffusionomega", "radiativediffusionouterboundary", "radiativediffusiontest1d",
    "radiativediffusiontest2d", "radiativediffusiontest2ddensity", "radiativediffusiontest2dk",
    "radiativediffusiontest2dsteps", "radiativediffusiontolerance", "randomfactor", "randomseed", "randomsigma",
    "rmax", "rmin", "rochelobeoverflow", "rofaveragingtime", "rofgamma", "rofplanet", "roframpingtime",
    "roftemperature", "rofvalue", "rofvariabletransfer", "scurvetype", "secondarydisk", "selfgravity",
    "selfgravityaspectratiochangethreshold", "selfgravitymode", "selfgravitystepsbetweenkernelupdate", "setsigma0",
    "shocktube", "sigma0", "sigmacondition", "sigmafilename", "sigmafloor", "sigmaslope", "spreadingring",
    "stabilizeviscosity", "surfacecooling", "t0", "taufactor", "taumin", "temp0", "temperature0",
    "thicknesssmoothing", "thicknesssmoothingsg", "transport", "vazimuthalconsidersquadropolemoment",
    "viscaccretmassflowtest", "viscousalpha", "viscousoutflowspeed", "writealpha", "writealphagrav",
    "writealphagravmean", "writealphareynolds", "writealphareynoldsmean", "writeaspectratio", "writeateverytimestep",
    "writedefaultvalues", "writedensity", "writediskquantities", "writedivv", "writeeccentricity",
    "writeeccentricitychange", "writeeffectivegamma", "writeenergy", "writefirstadiabaticindex", "writegastorques",
    "writekappa", "writelightcurves", "writelightcurvesradii", "writemassflow", "writemeanmolecularweight",
    "writepdv", "writepotential", "writepressure", "writeqminus", "writeqplus", "writeradialdissipation",
    "writeradialluminosity", "writescaleheight", "writesgaccelazi", "writesgaccelrad", "writesoundspeed", "writetau",
    "writetaucool", "writetemperature", "writetgravitational", "writetoomre", "writetorques", "writetreynolds",
    "writevelocity", "writeverticalopticaldepth", "writeviscosity", "writevisibility",};
// Features of the reference outside the gas path of this driver: a setup that switches one on is refused, not run
// without it (out of scope).
const char *const kRefusedWhenOn[] = {"selfgravity", "integrateparticles", "radiativediffusion", "rochelobeoverflow",
                                      "keepdiskmassconstant", "circumbinaryring", "planetorbitdisktest",
                                      "viscaccretmassflowtest"};

// code units (set_baseunits, src/units.cpp:131-185): l0 and m0 from the setup (default 1 au, 1 solMass), time and
// temperature units derived with G = 1 and R_gas = 1
const double G_CGS = 6.67430e-8, KB = 1.380649e-16, MU = 1.66053906660e-24;
const double AU_CM = 1.495978707e13, SOLMASS_G = 1.988409870698051e33;
double L0 = AU_CM, M0 = SOLMASS_G;
double TEMP0 = G_CGS * MU / KB * M0 / L0;
double TIME0 = std::sqrt(L0 * L0 * L0 / (G_CGS * M0));

// "<number> [unit]" -> code units for the quantity kind
enum Kind { K_NONE, K_LEN, K_MASS, K_SIGMA, K_TEMP, K_VISC };
double number(const std::string &s, Kind kind)
{
    std::istringstream is(s);
    double v = 0;
    std::string u;
    is >> v;
    is >> u;
    u = lower(u);
    if (u.empty())
        return v;
    if (kind == K_LEN) {
        if (u == "au")
            return v * AU_CM / L0;
        if (u == "cm")
            return v / L0;
        if (u == "solradius")
            return v * 6.957e10 / L0;
    } else if (kind == K_MASS) {
        if (u == "solmass")
            return v * SOLMASS_G / M0;
        if (u == "jupitermass")
            return v * 9.547919e-4 * SOLMASS_G / M0;
        if (u == "earthmass")
            return v * 3.0034893e-6 * SOLMASS_G / M0;
    } else if (kind == K_SIGMA) {
        if (u == "g/cm2")
            return v / (M0 / (L0 * L0));
    } else if (kind == K_TEMP) {
        if (u == "k")
            return v / TEMP0;
    } else if (kind == K_VISC) {
        if (u == "cm2/s")
            return v / (L0 * L0 / TIME0);
    }
    fprintf(stderr, "fargocpt_hip: unit '%s' in '%s' not understood\n", u.c_str(), s.c_str());
    exit(2);
}
double num(const Config &c, const std::string &k, double def, Kind kind = K_NONE)
{
    if (!c.has(k)) {
        c.used[lower(k)] = true;
        return def;
    }
    return number(c.str(k, ""), kind);
}

int bc_of(const std::string &s)
{
    const std::string v = lower(s);
    if (v == "zerogradient") return FCPT_BC_ZEROGRADIENT;
    if (v == "reference") return FCPT_BC_REFERENCE;
    if (v == "reflecting") return FCPT_BC_REFLECTING;
    if (v == "outflow") return FCPT_BC_OUTFLOW;
    if (v == "keplerian") return FCPT_BC_KEPLERIAN;
    if (v == "zeroshear") return FCPT_BC_ZEROSHEAR;
    if (v == "none") return FCPT_BC_NONE;
    fprintf(stderr, "fargocpt_hip: boundary condition '%s' is not supported\n", s.c_str());
    exit(2);
}
int damp_of(const std::string &s) // damping.cpp:152-178
{
    switch (s.empty() ? 'n' : std::tolower(s[0])) {
    case 'r': case 'i': case 'y': return FCPT_DAMP_REFERENCE;
    case 'm': return FCPT_DAMP_MEAN;
    case 'z': return FCPT_DAMP_ZERO;
    default: return FCPT_DAMP_NONE;
    }
}

// parameters::read + Interpret (src/parameters.cpp:520-900, src/Interpret.cpp:73-700) for the
// keys of the path
void config_to_desc(const Config &c, fcpt_desc &d)
{
    fcpt_desc_default(&d);
    { // base units: "l0: 30 au" / "m0: 1 solMass", or plain numbers in au / solMass (units.cpp:131-160)
        auto base = [&](const char *key, double unit_cgs, const char *unit_name) {
            std::istringstream is(c.str(key, "1.0"));
            double v = 1.0;
            std::string u;
            is >> v >> u;
            if (!u.empty() && lower(u) != unit_name) {
                fprintf(stderr, "fargocpt_hip: %s: unit '%s' not understood (%s)\n", key, u.c_str(), unit_name);
                exit(2);
            }
            return v * unit_cgs;
        };
        L0 = base("l0", AU_CM, "au");
        M0 = base("m0", SOLMASS_G, "solmass");
        TEMP0 = G_CGS * MU / KB * M0 / L0;
        TIME0 = std::sqrt(L0 * L0 * L0 / (G_CGS * M0));
        if (L0 != AU_CM || M0 != SOLMASS_G) {
            // the code-unit values of sigma_SB, c and the cgs factors of the opacity laws (constants.cpp:236-262), with
            // the expressions fcpt_desc_default uses for the default units
            const double h_cgs = 6.62607015e-27, c_cgs = 2.99792458e10;
            const double E0 = M0 * L0 * L0 / (TIME0 * TIME0);
            const double sigma_cgs = 2. * std::pow(M_PI, 5) * std::pow(KB, 4) / (15. * std::pow(h_cgs, 3) * std::pow(c_cgs, 2));
            d.temperature_cgs = TEMP0;
            d.density_cgs = M0 / (L0 * L0 * L0);
            d.opacity_cgs = L0 * L0 / M0;
            d.sigma_sb = sigma_cgs / (E0 / (L0 * L0 * TIME0 * TEMP0 * TEMP0 * TEMP0 * TEMP0));
            d.c_light = c_cgs / (L0 / TIME0);
        }
    }
    d.nr_global = (int)num(c, "Nrad", 64);
    d.nphi = (int)num(c, "Naz", 64);
    d.rmin = num(c, "Rmin", d.rmin, K_LEN);
    d.rmax = num(c, "Rmax", d.rmax, K_LEN);
    switch (std::tolower(c.str("RadialSpacing", "Arithmetic")[0])) {
    case 'l': d.radial_spacing = FCPT_SPACING_LOGARITHMIC; break;
    case 'e': d.radial_spacing = FCPT_SPACING_EXPONENTIAL; break;
    default: d.radial_spacing = FCPT_SPACING_ARITHMETIC;
    }
    d.exponential_cell_size_factor = num(c, "ExponentialCellSizeFactor", 1.41);
    const std::string eos = lower(c.str("EquationOfState", "Isothermal"));
    d.eos = (eos == "ideal" || eos == "adiabatic") ? FCPT_EOS_IDEAL : FCPT_EOS_ISOTHERMAL;
    d.adiabatic_index = num(c, "AdiabaticIndex", 7.0 / 5.0);
    if (d.eos == FCPT_EOS_IDEAL && d.adiabatic_index == 1.0)
        d.eos = FCPT_EOS_ISOTHERMAL; // Interpret.cpp:425-431
    d.mu = num(c, "mu", 1.0);
    d.aspect_ratio = num(c, "AspectRatio", 0.05);
    {
        const double t0 = num(c, "Temperature0", -1.0, K_TEMP); // Interpret.cpp:194-197
        if (t0 > 0.0)
            d.aspect_ratio = std::sqrt(t0 * d.Rgas / d.mu);
    }
    d.flaring_index = num(c, "FlaringIndex", 0.0);
    { // cps: cells per scale height overwrite Nrad and Naz (Interpret.cpp:206-228)
        const double cps = num(c, "cps", -1.0), H = d.aspect_ratio;
        if (cps > 0) {
            if (d.radial_spacing == FCPT_SPACING_ARITHMETIC) {
                d.nr_global = (int)std::round(cps * (d.rmax - d.rmin) / H);
                d.nphi = (int)std::round(2 * M_PI / (d.rmax - d.rmin) * d.nr_global);
            } else if (d.radial_spacing == FCPT_SPACING_LOGARITHMIC) {
                d.nr_global = (int)std::round(std::log(d.rmax / d.rmin) / std::log(1 + H / cps));
                d.nphi = (int)std::round(2 * M_PI / (std::pow(d.rmax / d.rmin, 1.0 / (double)d.nr_global) - 1));
            } else {
                fprintf(stderr, "fargocpt_hip: Setting resolution is not supported for the selected radial grid spacing.\n");
                exit(2);
            }
        }
    }
    d.minimum_temperature = num(c, "MinimumTemperature", 3.0 / TEMP0, K_TEMP);
    d.maximum_temperature = num(c, "MaximumTemperature", 1.0e300 / TEMP0, K_TEMP);
    d.sigma0 = num(c, "Sigma0", 173.0 / (M0 / (L0 * L0)), K_SIGMA);
    d.sigma_slope = num(c, "SigmaSlope", 0.0);
    d.sigma_floor = num(c, "SigmaFloor", 1e-9);
    d.set_sigma0 = c.flag("SetSigma0", false);
    d.disk_mass = num(c, "DiskMass", 0.01, K_MASS);
    d.viscous_alpha = num(c, "ViscousAlpha", 0.0);
    d.constant_viscosity = num(c, "ConstantViscosity", 0.0, K_VISC);
    d.radial_viscosity_factor = num(c, "RadialViscosityFactor", 1.0);
    d.stabilize_viscosity = (int)num(c, "StabilizeViscosity", 0);
    switch (std::tolower(c.str("ArtificialViscosity", "SN")[0])) {
    case 'n': d.artificial_viscosity = FCPT_ARTVISC_NONE; break;
    case 't': d.artificial_viscosity = FCPT_ARTVISC_TW; break;
    default: d.artificial_viscosity = FCPT_ARTVISC_SN;
    }
    d.artificial_viscosity_dissipation = c.flag("ArtificialViscosityDissipation", true);
    d.artificial_viscosity_factor = num(c, "ArtificialViscosityFactor", 1.41);
    d.heating_viscous = c.flag("HeatingViscous", true);
    d.heating_viscous_factor = num(c, "HeatingViscousFactor", 1.0);
    { // cooling terms (src/parameters.cpp:399-490,661-665)
        const std::string sc = lower(c.str("SurfaceCooling", "No"));
        if (sc == "thermal") {
            d.cooling_surface = 1;
        } else if (!(sc == "no" || sc == "off" || sc == "false")) {
            fprintf(stderr, "fargocpt_hip: SurfaceCooling: %s is not supported\n", sc.c_str());
            exit(2);
        }
        d.cooling_radiative_factor = num(c, "CoolingRadiativeFactor", 1.0);
        const std::string op = lower(c.str("Opacity", "Lin"));
        if (op == "lin") { // read_opacity_config, src/parameters.cpp:424-439
            d.opacity = FCPT_OPACITY_LIN;
        } else if (op == "bell") {
            d.opacity = FCPT_OPACITY_BELL;
        } else if (op == "constant") {
            d.opacity = FCPT_OPACITY_CONST;
        } else if (op == "simple") {
            d.opacity = FCPT_OPACITY_SIMPLE;
        } else {
            fprintf(stderr, "fargocpt_hip: Invalid choice for opacity type: %s\n", op.c_str());
            exit(2);
        }
        d.kappa_const = num(c, "KappaConst", 1.0);
        d.kappa_factor = num(c, "KappaFactor", 1.0);
        d.tau_factor = num(c, "TauFactor", 0.5);
        d.tau_min = num(c, "TauMin", 0.01);
        d.density_factor = num(c, "DensityFactor", std::sqrt(2.0 * M_PI));
        d.cooling_beta = c.flag("CoolingBetaLocal", false);
        d.cooling_beta_value = num(c, "CoolingBeta", 1.0);
        d.cooling_beta_ramp_up = num(c, "CoolingBetaRampUp", 0.0);
        const std::string br = lower(c.str("CoolingBetaReference", "Zero"));
        d.cooling_beta_reference = br == "reference" ? FCPT_BETAREF_REFERENCE
                                   : br == "model"   ? FCPT_BETAREF_MODEL
                                   : br == "floor"   ? FCPT_BETAREF_FLOOR
                                                     : FCPT_BETAREF_ZERO;
    }
    d.fast_transport = std::tolower(c.str("Transport", "Fast")[0]) == 'f';
    const std::string lim = lower(c.str("FluxLimiter", "VanLeer"));
    d.flux_limiter = (lim == "mc" || lim == "m") ? FCPT_LIMITER_MC : FCPT_LIMITER_VANLEER;
    d.integrator = std::tolower(c.str("Integrator", "Euler")[0]) == 'l' ? FCPT_INTEGRATOR_LEAPFROG : FCPT_INTEGRATOR_EULER;
    d.cfl = num(c, "CFL", 0.5);
    d.cfl_max_var = num(c, "CFLmaxVar", 1.1);
    d.first_dt = num(c, "FirstDT", 1e-9);
    d.heating_cooling_cfl_limit = num(c, "HeatingCoolingCFLlimit", 10.0);
    d.monitor_timestep = num(c, "MonitorTimestep", 1.0);
    d.nmonitor = (int)num(c, "Nmonitor", 10);
    d.nsnapshots = (int)num(c, "Nsnapshots", 1000);
    d.omega_frame = num(c, "OmegaFrame", 0.0);
    d.thickness_smoothing = num(c, "ThicknessSmoothing", 0.6);
    d.body_force_from_potential = c.flag("BodyForceFromPotential", true);
    d.initialize_vradial_zero = c.flag("InitializeVradialZero", false);
    d.initialize_pure_keplerian = c.flag("InitializePureKeplerian", false);
    if ((int)num(c, "ShockTube", 0) == 1) {
        d.ic = FCPT_IC_SHOCKTUBE;
        d.G = d.Rgas = 1.0; // init_shock_tube_test, src/init.cpp:507-512
    } else if (c.flag("SpreadingRing", false)) {
        d.ic = FCPT_IC_SPREADING_RING;
    }
    // boundary_conditions/config.cpp:345-440 composites, :97-343 per-variable overrides
    const char *side[2] = {"Inner", "Outer"};
    for (int s = 0; s < 2; ++s) {
        const std::string comp = lower(c.str(std::string(side[s]) + "Boundary", "individual"));
        std::string sig = "zerogradient", en = "zerogradient", vr = "zerogradient";
        if (comp == "outflow") vr = "outflow";
        else if (comp == "reflecting") vr = "reflecting";
        else if (comp == "reference") sig = en = vr = "reference";
        else if (comp != "zerogradient" && comp != "individual") {
            fprintf(stderr, "fargocpt_hip: %sBoundary: %s is not supported\n", side[s], comp.c_str());
            exit(2);
        }
        d.bc_sigma[s] = bc_of(c.str(std::string(side[s]) + "BoundarySigma", sig));
        d.bc_energy[s] = bc_of(c.str(std::string(side[s]) + "BoundaryEnergy", en));
        d.bc_vrad[s] = bc_of(c.str(std::string(side[s]) + "BoundaryVrad", vr));
        d.bc_vaz[s] = bc_of(c.str(std::string(side[s]) + "BoundaryVazi", "keplerian"));
        d.keplerian_vaz_factor[s] = num(c, std::string(side[s]) + "BoundaryVaziKeplerianFactor", 1.0);
        d.keplerian_vrad_factor[s] = num(c, std::string(side[s]) + "BoundaryVradKeplerianFactor", 0.1);
        d.damp_vrad[s] = damp_of(c.str(std::string("DampingVRadial") + side[s], "None"));
        d.damp_vaz[s] = damp_of(c.str(std::string("DampingVAzimuthal") + side[s], "None"));
        d.damp_sigma[s] = damp_of(c.str(std::string("DampingSurfaceDensity") + side[s], "None"));
        d.damp_energy[s] = damp_of(c.str(std::string("DampingEnergy") + side[s], "None"));
    }
    d.profile_cutoff_outer = c.flag("ProfileCutoffOuter", false) ? 1 : 0; // parameters.cpp:762-776
    d.profile_cutoff_point_outer = num(c, "ProfileCutoffPointOuter", 1.0e300, K_LEN);
    d.profile_cutoff_width_outer = num(c, "ProfileCutoffWidthOuter", 1.0, K_LEN);
    d.profile_cutoff_inner = c.flag("ProfileCutoffInner", false) ? 1 : 0;
    d.profile_cutoff_point_inner = num(c, "ProfileCutoffPointInner", 0.0, K_LEN);
    d.profile_cutoff_width_inner = num(c, "ProfileCutoffWidthInner", 1.0, K_LEN);
    d.damping = c.flag("Damping", false);
    d.damping_inner_limit = num(c, "DampingInnerLimit", 1.05);
    d.damping_outer_limit = num(c, "DampingOuterLimit", 0.95);
    d.damping_time_factor = num(c, "DampingTimeFactor", 1.0);
    d.damping_time_radius_outer = num(c, "DampingTimeRadiusOuter", d.rmax, K_LEN);
    d.write_massflow = c.flag("WriteMassFlow", false) ? 1 : 0; // parameters.cpp:345
    // hydro centre = the first body (HydroFrameCenter: primary)
    if (!c.nbody.empty() && c.nbody[0].count("mass"))
        d.hydro_center_mass = number(c.nbody[0].at("mass"), K_MASS);
}

struct Body {
    double a, m, phase, rsm_factor;
    double rampup; // "ramp-up time" in orbital periods (planet.cpp:166-179)
};

void mkdirs(const std::string &p)
{
    std::string acc;
    for (size_t i = 0; i < p.size(); ++i) {
        acc += p[i];
        if (p[i] == '/' || i + 1 == p.size())
            mkdir(acc.c_str(), 0755);
    }
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ != FCPT_OK) {                                                            \
            fprintf(stderr, "fargocpt_hip: %s failed (%d): %s\n", #call, rc_, fcpt_last_error()); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// This process's slab: rank, the rows of the global grid it holds and the window of them it writes
// (Zero_or_active / Max_or_active, src/split.cpp:66-78)
struct Slab {
    int rank = 0, nranks = 1;
    fcpt_split s;
    int nphi = 0, nr_global = 0;
    bool master() const { return rank == 0; }
    bool multi() const { return nranks > 1; }
    size_t local_count(bool vec) const { return (size_t)(s.nr + (vec ? 1 : 0)) * nphi; }
    // rows [lo, hi) of the local grid that write2D / write1D store, at global row imin + lo: the overlap rings are
    // stripped, and only the last slab writes the last interface row of a vector grid (polargrid.cpp:150-172)
    void window(bool vec, int &lo, int &hi) const
    {
        lo = s.zero_or_active;
        hi = s.max_or_active + ((vec && s.is_last) ? 1 : 0);
    }
};
Slab g_slab;
fcpt_ctx *g_ctx = nullptr;

void barrier()
{
    if (g_slab.multi())
        CHECK(fcpt_comm_barrier(g_ctx));
}

bool is_vector_field(int field) { return field == FCPT_F_VRAD || field == FCPT_F_VRAD0 || field == FCPT_F_MASSFLOW; }

// rank 0 creates / truncates the files of a collective write before the others open them
void create_file(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        fprintf(stderr, "fargocpt_hip: cannot create %s\n", path.c_str());
        exit(1);
    }
    fclose(f);
}

// write2D (src/polargrid.cpp:135-180): raw FP64, global row-major; every slab writes its window at
// (IMIN + Zero_or_active) * Nsec.  Collective: the file exists (create_file + barrier) before this is called.
void write_grid(fcpt_ctx *ctx, int field, const std::string &path)
{
    const bool vec = is_vector_field(field);
    std::vector<double> buf(g_slab.local_count(vec));
    CHECK(fcpt_download(ctx, field, buf.data()));
    int lo, hi;
    g_slab.window(vec, lo, hi);
    const int fd = open(path.c_str(), O_WRONLY);
    const size_t nb = (size_t)(hi - lo) * g_slab.nphi * sizeof(double);
    const off_t off = (off_t)(g_slab.s.imin + lo) * g_slab.nphi * sizeof(double);
    if (fd < 0 || pwrite(fd, buf.data() + (size_t)lo * g_slab.nphi, nb, off) != (ssize_t)nb) {
        fprintf(stderr, "fargocpt_hip: rank %d cannot write %s\n", g_slab.rank, path.c_str());
        exit(1);
    }
    close(fd);
}

// t_polargrid::read2D (src/polargrid.cpp:301-353): raw doubles, ring-major; every slab takes its rows
// IMIN .. IMIN + NRadial (overlap rings included) out of the global file
bool read_grid(fcpt_ctx *ctx, int field, const std::string &path, bool required)
{
    const bool vec = is_vector_field(field);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        if (required) {
            fprintf(stderr, "fargocpt_hip: cannot read %s\n", path.c_str());
            exit(1);
        }
        return false;
    }
    const size_t n_global = (size_t)(g_slab.nr_global + (vec ? 1 : 0)) * g_slab.nphi, n = g_slab.local_count(vec);
    fseek(f, 0, SEEK_END);
    const size_t have = (size_t)ftell(f) / sizeof(double);
    if (have != n_global) {
        fprintf(stderr, "fargocpt_hip: %s holds %zu values, expected %zu (Nrad / Naz changed?)\n", path.c_str(), have, n_global);
        exit(1);
    }
    std::vector<double> buf(n);
    fseek(f, (long)((size_t)g_slab.s.imin * g_slab.nphi * sizeof(double)), SEEK_SET);
    const size_t got = fread(buf.data(), sizeof(double), n, f);
    fclose(f);
    if (got != n) {
        fprintf(stderr, "fargocpt_hip: short read of %s\n", path.c_str());
        exit(1);
    }
    CHECK(fcpt_upload(ctx, field, buf.data()));
    return true;
}

struct misc_entry { // src/output.h:16-24
    unsigned int timestep;
    unsigned int nTimeStep;
    double time;
    double OmegaFrame;
    double FrameAngle;
    double last_dt;
    unsigned long int N_iter;
};

char **g_argv = nullptr;

// The parent of `--ranks n`: starts n copies of this program, one per slab, before anything here has touched a GPU
// (fork + exec at once), hands them FCPT_RANK / FCPT_NRANKS / FCPT_RDV (a fresh rendezvous directory under the output
// directory: the RCCL id or the shared-memory file of the host-staged transport), and waits for all of them.  The
// first rank that fails ends the others -- they would wait for it in a collective -- with SIGTERM to exactly the
// children started here.
int launch_ranks(int n, const std::string &outdir, bool quiet)
{
    mkdirs(outdir);
    const std::string rdv = outdir + ".fcpt_rdv_" + std::to_string((long)getpid());
    mkdir(rdv.c_str(), 0700);
    std::vector<pid_t> pids((size_t)n, (pid_t)-1);
    for (int r = 0; r < n; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("fargocpt_hip: fork");
            for (int k = 0; k < r; ++k)
                kill(pids[k], SIGTERM);
            return 1;
        }
        if (pid == 0) {
            setenv("FCPT_RANK", std::to_string(r).c_str(), 1);
            setenv("FCPT_NRANKS", std::to_string(n).c_str(), 1);
            setenv("FCPT_RDV", rdv.c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); // dmabuf IPC: what RCCL needs between processes on this driver
            execv("/proc/self/exe", g_argv);
            perror("fargocpt_hip: execv");
            _exit(127);
        }
        pids[r] = pid;
    }
    int failed = 0, alive = n;
    double t_term = 0.0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (alive > 0) {
        bool progress = false;
        for (int r = 0; r < n; ++r) {
            if (pids[r] < 0)
                continue;
            int st = 0;
            const pid_t w = waitpid(pids[r], &st, WNOHANG);
            if (w != pids[r])
                continue;
            progress = true;
            pids[r] = -1;
            --alive;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            if (rc != 0 && !failed) {
                failed = rc;
                fprintf(stderr, "fargocpt_hip: rank %d ended with code %d: ending the other ranks\n", r, rc);
                for (int k = 0; k < n; ++k)
                    if (pids[k] > 0)
                        kill(pids[k], SIGTERM);
                t_term = now();
            }
        }
        if (failed && alive > 0 && now() - t_term > 10.0) {
            for (int k = 0; k < n; ++k)
                if (pids[k] > 0)
                    kill(pids[k], SIGKILL);
            t_term = now() + 1e9;
        }
        if (!progress)
            usleep(2000);
    }
    for (const char *f : {"/rccl_id", "/rccl_id.tmp", "/hostlink"})
        unlink((rdv + f).c_str());
    rmdir(rdv.c_str());
    if (!quiet && !failed)
        printf("fargocpt_hip: %d ranks finished\n", n);
    return failed ? 1 : 0;
}

} // namespace

int main(int argc, char **argv)
{
    g_argv = argv;
    bool quiet = false, lenient = false;
    long max_steps = -1, restart_from = -1;
    int want_ranks = 1;
    std::string mode, cfgpath, transport = "auto";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-q")
            quiet = true;
        else if (a == "--lenient")
            lenient = true;
        else if ((a == "--ranks" || a == "-np") && i + 1 < argc)
            want_ranks = atoi(argv[++i]);
        else if (a == "--transport" && i + 1 < argc)
            transport = lower(argv[++i]);
        else if (a == "-N" && i + 1 < argc)
            max_steps = atol(argv[++i]);
        else if (mode.empty())
            mode = a;
        else if (mode == "restart" && restart_from < 0 && !a.empty() && a.find_first_not_of("0123456789") == std::string::npos)
            restart_from = atol(a.c_str());
        else
            cfgpath = a;
    }
    // start_mode.cpp:29-113: start | restart [N] | auto
    if ((mode != "start" && mode != "restart" && mode != "auto") || cfgpath.empty() || want_ranks < 1 ||
        (transport != "auto" && transport != "rccl" && transport != "host")) {
        fprintf(stderr, "usage: fargocpt_hip [-q] [--lenient] [-N steps] [--ranks n] [--transport auto|rccl|host] "
                        "start|auto|restart [N] <config.yml>\n");
        return 2;
    }
    // this process's rank: FCPT_RANK / FCPT_NRANKS from the --ranks parent below or from a launcher of the caller's
    Slab &slab = g_slab;
    const char *env_rank = getenv("FCPT_RANK"), *env_nranks = getenv("FCPT_NRANKS");
    if (env_rank && env_nranks) {
        slab.rank = atoi(env_rank);
        slab.nranks = atoi(env_nranks);
        if (slab.nranks < 1 || slab.rank < 0 || slab.rank >= slab.nranks) {
            fprintf(stderr, "fargocpt_hip: FCPT_RANK=%s FCPT_NRANKS=%s\n", env_rank, env_nranks);
            return 2;
        }
        if (want_ranks > 1 && want_ranks != slab.nranks) {
            fprintf(stderr, "fargocpt_hip: --ranks %d but FCPT_NRANKS=%d\n", want_ranks, slab.nranks);
            return 2;
        }
    }
    if (slab.rank > 0)
        quiet = true; // one voice, as logging::print_master
    Config cfg;
    if (!cfg.load(cfgpath)) {
        fprintf(stderr, "Can not find config file %s!\n", cfgpath.c_str());
        return 1;
    }
    fcpt_desc d;
    config_to_desc(cfg, d);
    std::string outdir = cfg.str("OutputDir", "output/out");
    if (outdir.back() != '/')
        outdir += "/";
    { // config::Config::exit_on_unknown_key (src/config.cpp:119-138, called at src/main.cpp:111-113)
        std::string unknown;
        for (auto &kv : cfg.kv) {
            bool known = false;
            for (const char *k : kReferenceKeys)
                known = known || kv.first == k;
            if (!known)
                unknown += (unknown.empty() ? "" : ", ") + kv.first;
        }
        if (!unknown.empty() && slab.rank == 0) {
            fprintf(stderr, "%sUnknown key(s) found in config file: '%s'\nMaybe there is a typo?\n",
                    lenient ? "fargocpt_hip: warning: " : "", unknown.c_str());
        }
        if (!unknown.empty() && !lenient)
            return 1;
        for (const char *k : kRefusedWhenOn)
            if (cfg.has(k) && cfg.flag(k, false)) {
                fprintf(stderr, "fargocpt_hip: '%s' is switched on, but that part of the reference is outside this driver's "
                                "gas path\n", k);
                return 2;
            }
    }
    if (want_ranks > 1 && !(env_rank && env_nranks))
        return launch_ranks(want_ranks, outdir, quiet); // the n ranks run, this process only waits for them
    // start_mode::configure_start_mode (src/start_mode.cpp:29-113): auto = restart from the last snapshot of
    // snapshots/list.txt if there is one; restart without a number likewise
    std::string restart_dir;
    if (mode == "auto" || mode == "restart") {
        std::string id = restart_from >= 0 ? std::to_string(restart_from) : "";
        if (id.empty()) {
            std::ifstream list(outdir + "snapshots/list.txt");
            std::string line;
            while (std::getline(list, line))
                if (!line.empty())
                    id = line; // output::get_last_snapshot_id
        }
        FILE *mf = id.empty() ? nullptr : fopen((outdir + "snapshots/" + id + "/misc.bin").c_str(), "rb");
        if (mf) {
            fclose(mf);
            restart_dir = outdir + "snapshots/" + id + "/";
            mode = "restart";
        } else if (mode == "restart" && restart_from >= 0) {
            fprintf(stderr, "fargocpt_hip: cannot read %ssnapshots/%s/misc.bin\n", outdir.c_str(), id.c_str());
            return 1;
        } else {
            if (!quiet)
                printf("No output found, starting fresh simulation\n");
            mode = "start";
        }
    }
    const bool restarting = mode == "restart";
    // bodies: star + planets on fixed circular orbits
    std::vector<Body> bodies;
    for (size_t k = 0; k < cfg.nbody.size(); ++k) {
        const auto &b = cfg.nbody[k];
        Body o;
        o.a = b.count("semi-major axis") ? number(b.at("semi-major axis"), K_LEN) : 0.0;
        o.m = b.count("mass") ? number(b.at("mass"), K_MASS) : 0.0;
        o.phase = 0.0;
        o.rsm_factor = b.count("cubic smoothing factor") ? number(b.at("cubic smoothing factor"), K_NONE) : 0.0;
        o.rampup = b.count("ramp-up time") ? number(b.at("ramp-up time"), K_NONE) : 0.0;
        bodies.push_back(o);
    }
    if (bodies.empty())
        bodies.push_back({0.0, d.hydro_center_mass, 0.0, 0.0, 0.0});

    std::vector<double> radii(d.nr_global + FCPT_GEOM_PAD + 1);
    CHECK(fcpt_radii(&d, radii.data()));
    // The initial fields of the GLOBAL grid (rank 0 of 1), of which every slab then takes its rows IMIN .. IMIN + NRadial:
    // the reference fills each rank's rows from the same formulas of the global ring index, and the one global quantity
    // of init_physics -- SetSigma0's disk mass, an MPI sum there -- is then the same number on every rank by construction
    const size_t ns = (size_t)d.nr_global * d.nphi, nv = (size_t)(d.nr_global + 1) * d.nphi;
    std::vector<double> sigma(ns), vrad(nv), vazi(ns), energy(ns);
    d.rank = 0;
    d.nranks = 1;
    CHECK(fcpt_initial_fields(&d, radii.data(), sigma.data(), vrad.data(), vazi.data(), energy.data()));
    // SigmaCondition / EnergyCondition: 2D (init.cpp:1002-1007,1307-1312: read2D of the named file); 1D needs the
    // reference's GSL spline and is not offered
    for (int q = 0; q < 2; ++q) {
        const std::string cond = lower(cfg.str(q == 0 ? "SigmaCondition" : "EnergyCondition", "profile"));
        const std::string fn = cfg.str(q == 0 ? "SigmaFilename" : "EnergyFilename", "");
        if (cond == "profile" || cond.empty())
            continue;
        if (cond != "2d" || (q == 1 && d.eos != FCPT_EOS_IDEAL)) {
            if (q == 1 && cond == "2d")
                continue; // no energy equation
            fprintf(stderr, "fargocpt_hip: %sCondition: %s is not supported (Profile, 2D)\n", q == 0 ? "Sigma" : "Energy", cond.c_str());
            return 2;
        }
        if (d.set_sigma0 || d.profile_cutoff_inner || d.profile_cutoff_outer) {
            fprintf(stderr, "fargocpt_hip: %sCondition: 2D together with SetSigma0 or ProfileCutoff* is not supported\n", q == 0 ? "Sigma" : "Energy");
            return 2;
        }
        std::vector<double> &dst = q == 0 ? sigma : energy;
        FILE *f = fopen(fn.c_str(), "rb");
        if (!f || fread(dst.data(), sizeof(double), ns, f) != ns) {
            fprintf(stderr, "fargocpt_hip: cannot read %zu values from '%s'\n", ns, fn.c_str());
            return 1;
        }
        fclose(f);
    }
    // ---- this rank's slab, its GPU and its communicator (SplitDomain, src/split.cpp:34-88; src/parallel.cpp:28-40) ------
    d.rank = slab.rank;
    d.nranks = slab.nranks;
    CHECK(fcpt_split_domain(&d, &slab.s));
    slab.nphi = d.nphi;
    slab.nr_global = d.nr_global;
    int32_t ndev = 0;
    CHECK(fcpt_device_count(&ndev));
    if (ndev <= 0) {
        fprintf(stderr, "fargocpt_hip: no HIP device: the gas update has no CPU path\n");
        return 1;
    }
    if (transport == "auto")
        transport = slab.nranks <= ndev ? "rccl" : "host";
    if (slab.multi() && transport == "rccl" && slab.nranks > ndev) {
        fprintf(stderr, "fargocpt_hip: --transport rccl needs one GPU per rank (%d ranks, %d GPU(s)): RCCL refuses two "
                        "ranks on one device\n", slab.nranks, (int)ndev);
        return 2;
    }
    CHECK(fcpt_set_device(slab.rank % ndev));
    fcpt_ctx *ctx = nullptr;
    CHECK(fcpt_create(&d, radii.data(), &ctx));
    g_ctx = ctx;
    {
        const size_t row0 = (size_t)slab.s.imin * d.nphi; // local row 0 = global row IMIN
        CHECK(fcpt_upload(ctx, FCPT_F_SIGMA, sigma.data() + row0));
        CHECK(fcpt_upload(ctx, FCPT_F_VRAD, vrad.data() + row0));
        CHECK(fcpt_upload(ctx, FCPT_F_VAZI, vazi.data() + row0));
        CHECK(fcpt_upload(ctx, FCPT_F_ENERGY, energy.data() + row0));
        for (std::vector<double> *v : {&sigma, &vrad, &vazi, &energy})
            std::vector<double>().swap(*v); // the global copies are not needed any more
    }
    if (slab.multi()) {
        const char *e = getenv("FCPT_RDV");
        const std::string rdv = e && e[0] ? std::string(e) : outdir + ".fcpt_rdv";
        if (slab.master())
            mkdirs(rdv + "/");
        if (transport == "host") {
            CHECK(fcpt_comm_init_host(ctx, (rdv + "/hostlink").c_str()));
        } else {
            // ncclGetUniqueId on slab 0; the 128 bytes travel by file (the reference's host would MPI_Bcast them)
            unsigned char id[FCPT_COMM_ID_BYTES];
            const std::string idf = rdv + "/rccl_id";
            if (slab.master()) {
                CHECK(fcpt_comm_unique_id(id));
                FILE *f = fopen((idf + ".tmp").c_str(), "wb");
                if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id) || fclose(f) != 0 || rename((idf + ".tmp").c_str(), idf.c_str()) != 0) {
                    fprintf(stderr, "fargocpt_hip: cannot write %s\n", idf.c_str());
                    return 1;
                }
            } else {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    FILE *f = fopen(idf.c_str(), "rb");
                    const bool ok = f && fread(id, 1, sizeof(id), f) == sizeof(id);
                    if (f)
                        fclose(f);
                    if (ok)
                        break;
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                        fprintf(stderr, "fargocpt_hip: rank %d: no RCCL id in %s after 120 s\n", slab.rank, idf.c_str());
                        return 1;
                    }
                    usleep(2000);
                }
            }
            CHECK(fcpt_comm_init(ctx, id));
        }
        if (!quiet)
            printf("fargocpt_hip: %d radial slabs, %s transport, %d GPU(s) visible\n", slab.nranks,
                   transport == "host" ? "host-staged" : "RCCL", (int)ndev);
    }

    // Bodies at time t for the potential of a step of length dt (CalculateNbodyPotential, Pframeforce.cpp:21-94):
    // positions on the circular orbits, masses ramped up (get_rampup_mass, planet.cpp:166-179), and the indirect term
    // of the star-centred frame without the disk (refframe::IndirectTermPlanets): minus the acceleration of the star,
    // which the reference takes as the velocity change of the hydro centre over the step divided by dt
    // (ComputeIndirectTermNbody, frame_of_reference.cpp:138-160) -- for circular orbits the time average of
    // G m r_p / a^3 over [t, t + dt], in closed form.
    const bool indirect = lower(cfg.str("HydroFrameCenter", "primary")) == "primary";
    auto set_bodies = [&](double t, double dt) {
        double x[FCPT_MAX_BODIES], y[FCPT_MAX_BODIES], m[FCPT_MAX_BODIES], rsm[FCPT_MAX_BODIES];
        const int n = (int)std::min<size_t>(bodies.size(), FCPT_MAX_BODIES);
        double itx = 0.0, ity = 0.0;
        for (int k = 0; k < n; ++k) {
            const Body &b = bodies[k];
            const double om = b.a > 0 ? std::sqrt(d.G * (d.hydro_center_mass + b.m) / (b.a * b.a * b.a)) : 0.0;
            const double ang = b.phase + (om - d.omega_frame) * t;
            x[k] = b.a * std::cos(ang);
            y[k] = b.a * std::sin(ang);
            m[k] = b.m;
            if (b.rampup > 0 && om > 0) {
                const double period = 2 * M_PI / om;
                if (t < b.rampup * period) {
                    const double cs = std::cos(t * M_PI_2 / (b.rampup * period));
                    m[k] = b.m * (1.0 - cs * cs);
                }
            }
            // dimensionless Roche radius ~ (q/3)^(1/3) (Theo.cpp:251-277 converges to it for small q)
            rsm[k] = b.a * std::cbrt(b.m / (3.0 * d.hydro_center_mass)) * b.rsm_factor;
            if (indirect && k > 0 && b.a > 0 && dt > 0) {
                const double g = d.G * b.m / (b.a * b.a), w = om * dt;
                itx -= g * (std::sin(ang + w) - std::sin(ang)) / w;
                ity -= g * (std::cos(ang) - std::cos(ang + w)) / w;
            }
        }
        CHECK(fcpt_set_bodies(ctx, n, x, y, m, rsm, itx, ity));
    };
    set_bodies(0.0, 0.0);
    { // irradiating bodies: 'temperature', 'radius', 'irradiation ramp-up time' (planetary_system.cpp:160-250)
        double temp[FCPT_MAX_BODIES] = {0}, rad[FCPT_MAX_BODIES] = {0}, ramp[FCPT_MAX_BODIES] = {0};
        bool any = false;
        const int n = (int)std::min<size_t>(cfg.nbody.size(), FCPT_MAX_BODIES);
        for (int k = 0; k < n; ++k) {
            const auto &b = cfg.nbody[k];
            temp[k] = b.count("temperature") ? number(b.at("temperature"), K_TEMP) : 0.0;
            rad[k] = b.count("radius") ? number(b.at("radius"), K_LEN) : 0.0;
            ramp[k] = b.count("irradiation ramp-up time") ? number(b.at("irradiation ramp-up time"), K_NONE) : 0.0;
            any = any || temp[k] > 0.0;
        }
        if (any && d.eos == FCPT_EOS_IDEAL) // (without the energy equation a body's temperature heats nothing)
            CHECK(fcpt_set_body_irradiation(ctx, n, temp, rad, ramp));
    }
    CHECK(fcpt_init_physics(ctx));

    // ---- output files -------------------------------------------------------------------------
    if (slab.master()) {
        mkdirs(outdir + "snapshots/");
        mkdirs(outdir + "monitor/");
    }
    if (!restarting && slab.master()) {
        FILE *f = fopen((outdir + "used_rad.dat").c_str(), "w"); // src/init.cpp:228-246
        for (int n = 0; n <= d.nr_global; ++n)
            fprintf(f, "%.18g\n", radii[n]);
        fclose(f);
        f = fopen((outdir + "dimensions.dat").c_str(), "w"); // src/parameters.cpp:1127-1176
        const char *sp[] = {"Arithmetic", "Logarithmic", "Exponential"};
        fprintf(f, "#RMIN\tRMAX\tPHIMIN\tPHIMAX          \tNRAD\tNAZ\tNGHRAD\tNGHAZ\tRadial_spacing\n");
        fprintf(f, "%.16g\t%.16g\t%.16g\t%.16g\t%d\t%d\t%d\t%d\t%s\n", d.rmin, d.rmax, 0.0, 2 * M_PI, d.nr_global, d.nphi, 1,
                1, sp[d.radial_spacing]);
        fclose(f);
        remove((outdir + "snapshots/list.txt").c_str());
        // units.yml (write_code_units_file, src/units.cpp:449-497) and info2D.yml (src/output.cpp:786-846):
        // what the reference's Python loader (python_module/fargocpt/data.py) needs beside the grids
        const double len = L0, mass = M0, tim = TIME0, temp = TEMP0;
        const double energy = len * len * mass / (tim * tim), vel = len / tim;
        struct U {
            const char *name, *sym;
            double v;
        };
        const U units[] = {
            {"length", "cm", len},
            {"mass", "g", mass},
            {"time", "s", tim},
            {"temperature", "K", temp},
            {"energy", "erg", energy},
            {"energy surface density", "erg cm^-2", mass / (tim * tim)},
            {"density", "g cm^-3", mass / (len * len * len)},
            {"mass surface density", "g cm^-2", mass / (len * len)},
            {"opacity", "g^-1 cm^2", len * len / mass},
            {"energy flux", "erg cm^-2 s^-1", energy / (len * len * tim)},
            {"velocity", "cm s^-1", vel},
            {"angular momentum", "cm^2 g s^-1", len * mass * vel},
            {"kinematic viscosity", "cm^2 s^-1", len * len / tim},
            {"dynamic viscosity", "P", mass / (len * tim)},
            {"acceleration", "cm s^-2", len / (tim * tim)},
            {"stress", "g s^-2", mass / (tim * tim)},
            {"pressure", "dyn cm^-1", mass / (tim * tim)},
            {"power", "erg/s", mass * len * len / (tim * tim * tim)},
            {"potential", "erg/g", len * len / (tim * tim)},
            {"torque", "erg", len * len * mass / (tim * tim)},
            {"force", "dyn", mass * len / (tim * tim)},
            {"mass accretion rate", "g s^-1", mass / tim},
        };
        f = fopen((outdir + "units.yml").c_str(), "w");
        fprintf(f, "# code units file\n# version 0.2\n\n");
        for (const U &u : units)
            fprintf(f, "%s:\n  cgs symbol: %s\n  cgs value: %.17g\n  unit: %.17g %s\n\n", u.name, u.sym, u.v, u.v, u.sym);
        fclose(f);
        struct G {
            const char *name, *sym;
            double v;
            bool vec;
        };
        const G grids[] = {{"Sigma", "g cm^-2", mass / (len * len), false},
                           {"vrad", "cm s^-1", vel, true},
                           {"vazi", "cm s^-1", vel, false},
                           {"energy", "erg cm^-2", mass / (tim * tim), false},
                           {"Temperature", "K", temp, false}};
        f = fopen((outdir + "info2D.yml").c_str(), "w");
        fprintf(f, "# 2D output variable descriptions\n# version 0.1\n\n");
        for (const G &g : grids) {
            if ((!strcmp(g.name, "energy") || !strcmp(g.name, "Temperature")) && d.eos != FCPT_EOS_IDEAL)
                continue;
            fprintf(f, "%s:\n  cgs symbols: %s\n  code_to_cgs_factor: %.17g\n  unit: %.17g %s\n  Nrad: %d\n  Nazi: %d\n",
                    g.name, g.sym, g.v, g.v, g.sym, g.vec ? d.nr_global + 1 : d.nr_global, d.nphi);
            fprintf(f, "  bigendian: 0\n  on_radial_interface: %s\n  on_azimuthal_interface: %s\n  filename: %s.dat\n\n",
                    g.vec ? "true" : "false", !strcmp(g.name, "vazi") ? "true" : "false", g.name);
        }
        fclose(f);
    }
    // boundary_conditions::initial_values_needed(): the t = 0 grids are part of the state (damping towards them,
    // reference boundary conditions)
    bool needs_reference = d.damping != 0;
    for (int sd = 0; sd < 2; ++sd)
        needs_reference = needs_reference || d.bc_sigma[sd] == FCPT_BC_REFERENCE || d.bc_energy[sd] == FCPT_BC_REFERENCE ||
                          d.bc_vrad[sd] == FCPT_BC_REFERENCE || d.bc_vaz[sd] == FCPT_BC_REFERENCE;
    std::function<void(unsigned, unsigned, const char *)> write_snapshot_as;
    auto write_snapshot = [&](unsigned nsnap, unsigned nmon) {
        write_snapshot_as(nsnap, nmon, nullptr);
        if (nsnap == 0 && needs_reference) // simulation.cpp:41-47: the damping data as a reference
            write_snapshot_as(nsnap, nmon, "reference");
    };
    write_snapshot_as = [&](unsigned nsnap, unsigned nmon, const char *name) {
        fcpt_clock clk;
        CHECK(fcpt_get_clock(ctx, &clk));
        const std::string dir = outdir + "snapshots/" + (name ? std::string(name) : std::to_string(nsnap)) + "/";
        std::vector<std::pair<int, std::string>> grids = {{FCPT_F_SIGMA, "Sigma.dat"}, {FCPT_F_VRAD, "vrad.dat"}, {FCPT_F_VAZI, "vazi.dat"}};
        if (d.eos == FCPT_EOS_IDEAL) {
            grids.push_back({FCPT_F_ENERGY, "energy.dat"});
            grids.push_back({FCPT_F_TEMPERATURE, "Temperature.dat"});
            // the CFL condition of the next step reads Q+ and Q- of the last one (cfl.cpp:303-316): restart.cpp:78-95
            grids.push_back({FCPT_F_QPLUS, "Qplus.dat"});
            grids.push_back({FCPT_F_QMINUS, "Qminus.dat"});
        }
        const bool massflow = d.write_massflow && !name;
        // MPI_File_open(MPI_MODE_CREATE) is collective in the reference; here rank 0 creates the directory and the
        // empty files, then every slab writes its window into them
        if (slab.master()) {
            mkdirs(dir);
            for (auto &g : grids)
                create_file(dir + g.second);
            if (massflow)
                create_file(dir + "MassFlow1D.dat");
        }
        barrier();
        for (auto &g : grids)
            write_grid(ctx, g.first, dir + g.second);
        if (massflow) {
            // MassFlow1D.dat (t_polargrid::write1D, polargrid.cpp:187-278, of a vector grid that is integrated over
            // azimuth): pairs (Rinf[n], sum_j MASSFLOW(n, j) / (Nmonitor MonitorTimestep)) -- calculate_massflow,
            // quantities.cpp:770-781 -- every slab its window of interfaces at (IMIN + Zero_or_active) * 2, then the
            // grid is cleared (data.cpp:277)
            std::vector<double> mf(slab.local_count(true));
            CHECK(fcpt_download(ctx, FCPT_F_MASSFLOW, mf.data()));
            int lo, hi;
            slab.window(true, lo, hi);
            std::vector<double> out(2 * (size_t)(hi - lo));
            const double denom = (double)d.nmonitor * d.monitor_timestep;
            for (int n = lo; n < hi; ++n) {
                double sum = 0.0;
                for (int j = 0; j < d.nphi; ++j)
                    sum += mf[(size_t)n * d.nphi + j] / denom;
                out[2 * (size_t)(n - lo)] = radii[slab.s.imin + n];
                out[2 * (size_t)(n - lo) + 1] = sum;
            }
            const int fd = open((dir + "MassFlow1D.dat").c_str(), O_WRONLY);
            const size_t nb = out.size() * sizeof(double);
            if (fd < 0 || pwrite(fd, out.data(), nb, (off_t)(slab.s.imin + lo) * 2 * sizeof(double)) != (ssize_t)nb) {
                fprintf(stderr, "fargocpt_hip: cannot write %sMassFlow1D.dat\n", dir.c_str());
                exit(1);
            }
            close(fd);
            std::fill(mf.begin(), mf.end(), 0.0);
            CHECK(fcpt_upload(ctx, FCPT_F_MASSFLOW, mf.data()));
        }
        barrier(); // the grids are complete before misc.bin and list.txt announce the snapshot
        if (!slab.master())
            return;
        misc_entry misc;
        memset(&misc, 0, sizeof(misc));
        misc.timestep = nsnap;
        misc.nTimeStep = nmon;
        misc.time = clk.time;
        misc.OmegaFrame = d.omega_frame;
        misc.last_dt = clk.last_dt;
        misc.N_iter = clk.n_hydro_iter;
        FILE *f = fopen((dir + "misc.bin").c_str(), "wb"); // src/output.cpp:494-527
        fwrite(&misc, sizeof(misc), 1, f);
        fclose(f);
        std::ifstream src(cfgpath, std::ios::binary);
        std::ofstream dst(dir + "config.yml", std::ios::binary);
        dst << src.rdbuf();
        if (name)
            return; // write_full_output(data, "reference", false): not registered
        f = fopen((outdir + "snapshots/list.txt").c_str(), "a");
        fprintf(f, "%u\n", nsnap);
        fclose(f);
        f = fopen((outdir + "snapshots/timeSnapshot.dat").c_str(), nsnap == 0 ? "w" : "a");
        if (nsnap == 0)
            fprintf(f, "# Time log for course output.\n#version: 0.1\n#variable: 0 | snapshot number | 1\n"
                       "#variable: 1 | monitor number | 1\n#variable: 2 | time | code\n");
        fprintf(f, "%u\t%u\t%#.16e\n", nsnap, nmon, clk.time);
        fclose(f);
        if (!quiet)
            printf("Writing output %s, Snapshot Number %u, Time %f.\n", dir.c_str(), nsnap, clk.time);
    };

    if (!quiet) // keys the reference knows and this driver's gas path does not consume
        for (auto &kv : cfg.kv)
            if (!cfg.used.count(kv.first) && kv.first.compare(0, 5, "write") != 0 && kv.first.compare(0, 8, "particle") != 0)
                fprintf(stderr, "fargocpt_hip: note: key '%s' is not used by the gas path\n", kv.first.c_str());

    // ---- main.cpp:117-152 and sim::run (simulation.cpp:505-558) --------------------------------
    auto calc_dt = [&]() { // condition_cfl incl. its MPI_Allreduce(MIN) over the slabs (cfl.cpp:379) + CalculateTimeStep
        double cfl, dt;
        if (slab.multi())
            CHECK(fcpt_cfl_allreduce(ctx, &cfl));
        else
            CHECK(fcpt_cfl(ctx, &cfl));
        CHECK(fcpt_calculate_timestep(ctx, cfl, &dt));
        return dt;
    };
    auto exchange = [&]() { // CommunicateBoundaries (commbound.cpp:98-182): returns at once for a single slab
        if (slab.multi())
            CHECK(fcpt_exchange(ctx));
    };
    calc_dt();          // main.cpp:117
    unsigned n_monitor = 0;
    unsigned long n_iter = 0, n_iter_last = 0;
    double time = 0.0;
    if (restarting) { // restart_load, src/restart.cpp:18-139
        misc_entry misc;
        FILE *mf = fopen((restart_dir + "misc.bin").c_str(), "rb");
        if (!mf || fread(&misc, sizeof(misc), 1, mf) != 1) {
            fprintf(stderr, "fargocpt_hip: cannot read %smisc.bin\n", restart_dir.c_str());
            return 1;
        }
        fclose(mf);
        if (needs_reference) {
            const std::string ref = outdir + "snapshots/reference/";
            read_grid(ctx, FCPT_F_SIGMA0, ref + "Sigma.dat", true);
            read_grid(ctx, FCPT_F_VRAD0, ref + "vrad.dat", true);
            read_grid(ctx, FCPT_F_VAZI0, ref + "vazi.dat", true);
            if (d.eos == FCPT_EOS_IDEAL)
                read_grid(ctx, FCPT_F_ENERGY0, ref + "energy.dat", true);
        }
        read_grid(ctx, FCPT_F_SIGMA, restart_dir + "Sigma.dat", true);
        read_grid(ctx, FCPT_F_VRAD, restart_dir + "vrad.dat", true);
        read_grid(ctx, FCPT_F_VAZI, restart_dir + "vazi.dat", true);
        if (d.eos == FCPT_EOS_IDEAL) {
            read_grid(ctx, FCPT_F_ENERGY, restart_dir + "energy.dat", true);
            const bool qp = read_grid(ctx, FCPT_F_QPLUS, restart_dir + "Qplus.dat", false);
            const bool qm = read_grid(ctx, FCPT_F_QMINUS, restart_dir + "Qminus.dat", false);
            if (!(qp && qm) && !quiet)
                printf("Cannot read Qplus / Qminus, no bitwise identical restarting possible!\n");
        }
        fcpt_clock clk;
        CHECK(fcpt_get_clock(ctx, &clk));
        clk.time = misc.time;
        clk.last_dt = misc.last_dt;
        clk.n_hydro_iter = misc.N_iter;
        clk.n_monitor = misc.nTimeStep;
        clk.n_snapshot = misc.timestep;
        CHECK(fcpt_set_clock(ctx, &clk));
        time = misc.time;
        n_monitor = misc.nTimeStep;
        n_iter = n_iter_last = misc.N_iter;
        set_bodies(time, 0.0);
        CHECK(fcpt_recalculate_derived(ctx));
        if (!quiet)
            printf("Restarting from %s at time %f (snapshot %u, monitor step %u).\n", restart_dir.c_str(), time,
                   misc.timestep, misc.nTimeStep);
    }
    exchange(); // CommunicateBoundariesAll, main.cpp:147
    if (!restarting)
        write_snapshot(0, 0); // main.cpp:150-152
    CHECK(fcpt_apply_boundary(ctx, 0.0, 0)); // sim::init (simulation.cpp:462-474)
    if (!restarting)
        calc_dt();
    exchange();

    const double t_final = (double)d.nsnapshots * d.nmonitor * d.monitor_timestep;
    FILE *tlog = slab.master() ? fopen((outdir + "monitor/timestepLogging.dat").c_str(), restarting ? "a" : "w") : nullptr;
    if (tlog && !restarting)
        fprintf(tlog, "#version: 2\n#FargoCPT Time log for the hydro timestep size.\n"
                  "#variable: 0  | snapshot number | 1\n#variable: 1  | monitor number | 1\n"
                  "#variable: 2  | hydrostep number | 1\n#variable: 3  | Number of Hydrosteps in last monitor_timestep | 1\n"
                  "#variable: 4  | time | code\n#variable: 5  | walltime | s\n#variable: 6  | walltime per hydrostep | ms\n"
                  "#variable: 7  | mean dt | code\n#variable: 8  | min dt | code\n#variable: 9  | max dt | code\n");
    double sum_dt = 0, min_dt = 1e300, max_dt = 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto t_last = t_start;
    const bool moving = bodies.size() > 1;
    // Between two monitor times the loop of sim::run does nothing but step: as long as the monitor-time snapping
    // cannot trigger (simulation.cpp:528-540: it needs time_left < 1.05 dt, and dt grows by at most CFLmaxVar per
    // step) those steps are exactly the ones fcpt_run_steps takes with dt resident on the device -- no read-back of
    // the CFL value per step, and a replayed hipGraph on the small grids of the reference's tests.  Bodies that move
    // need the host every step.
    double last_dt = 0.0;
    {
        fcpt_clock clk0;
        CHECK(fcpt_get_clock(ctx, &clk0));
        last_dt = clk0.last_dt;
    }
    CHECK(fcpt_dt_statistics(ctx, nullptr, nullptr, 1));
    while (time < t_final) {
        if (max_steps >= 0 && (long)n_iter >= max_steps)
            break;
        if (!moving && last_dt > 0.0) {
            const double left = (n_monitor + 1) * d.monitor_timestep - time;
            long k = 0;
            double used = 0.0, dtj = last_dt;
            for (; k < 100000; ++k) { // worst case: every step grows by CFLmaxVar
                dtj *= d.cfl_max_var;
                if (!(left - used >= 1.05 * dtj * (1.0 + 1e-9)))
                    break;
                used += dtj;
            }
            k -= 1; // one step of margin
            if (max_steps >= 0)
                k = std::min<long>(k, max_steps - (long)n_iter);
            if (k >= 8) {
                int64_t done = 0;
                CHECK(fcpt_run_steps(ctx, k, 0, &done));
                fcpt_clock clk;
                CHECK(fcpt_get_clock(ctx, &clk));
                sum_dt += clk.time - time;
                time = clk.time;
                last_dt = clk.last_dt;
                n_iter += (unsigned long)done;
                continue;
            }
        }
        const double cfl_dt = calc_dt();
        last_dt = cfl_dt;
        double step_dt;
        CHECK(fcpt_snap_to_monitor(ctx, cfl_dt, &step_dt));
        const double time_next_monitor = (n_monitor + 1) * d.monitor_timestep;
        if (moving)
            set_bodies(time, step_dt);
        CHECK(fcpt_step(ctx, step_dt));
        exchange(); // simulation.cpp:236
        CHECK(fcpt_post(ctx, step_dt));
        time += step_dt;
        ++n_iter;
        sum_dt += step_dt;
        if (std::fabs(time_next_monitor - time) < 1e-6 * cfl_dt) {
            ++n_monitor;
            fcpt_clock clk;
            CHECK(fcpt_get_clock(ctx, &clk));
            clk.n_monitor = n_monitor;
            clk.n_snapshot = n_monitor / (unsigned)d.nmonitor;
            CHECK(fcpt_set_clock(ctx, &clk));
            const auto now = std::chrono::steady_clock::now();
            const double wall = std::chrono::duration<double>(now - t_start).count();
            const unsigned long nint = n_iter - n_iter_last;
            CHECK(fcpt_dt_statistics(ctx, &min_dt, &max_dt, 1)); // hydro_dt_logger: kept next to the device clock
            const double ms = nint ? 1e3 * std::chrono::duration<double>(now - t_last).count() / nint : 0.0;
            if (tlog) {
                fprintf(tlog, "%u\t%u\t%lu\t%lu\t%#.16e\t%#.16e\t%#.16e\t%#.16e\t%#.16e\t%#.16e\n", clk.n_snapshot,
                        n_monitor, n_iter, nint, time, wall, ms, nint ? sum_dt / nint : 0.0, min_dt, max_dt);
                fflush(tlog);
            }
            t_last = now;
            n_iter_last = n_iter;
            sum_dt = 0;
            min_dt = 1e300;
            max_dt = 0;
            if (n_monitor % (unsigned)d.nmonitor == 0) // handle_outputs, simulation.cpp:50-66
                write_snapshot(n_monitor / (unsigned)d.nmonitor, n_monitor);
        }
    }
    if (tlog)
        fclose(tlog);
    CHECK(fcpt_synchronize(ctx));
    barrier(); // no slab tears its communicator down while another still waits in it
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (!quiet)
        printf("-- Final: Total Hydrosteps %lu, Time %.2f, Walltime %.2f seconds, Time per Step: %.2f milliseconds\n", n_iter,
               time, wall, n_iter ? 1e3 * wall / n_iter : 0.0);
    fcpt_destroy(ctx);
    return 0;
}
