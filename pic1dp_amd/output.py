"""pic1dp.out in the reference's byte layout, and the stdout progress lines.

The reference writes its diagnostics through a PETSc binary viewer, i.e.
big-endian, PetscInt = int32, PetscReal/PetscScalar = float64
(src/pic1dp_output.F90:74-92 header, :173-186 field record, :457-474
distribution record; reader tools/OutputData.py:26-79).  A file written here is
readable by the reference's own tools (visual.py, runinfo.py):

  header   int32[6 + nmode]   nspecies, nmode, nx, nv, nx_opd, nv_opd, modes(:)
           float64[2]         lx, v_max
  record   float64[2 + 3*nspecies]  time, int E^2 dx, per species: sum v^2,
                                    total and perturbed kinetic sums
           Vec field_mode_re, field_mode_im      (int32 1211214, int32 n, float64[n])
           Vec field_electric, field_chargeden
           per species: float64[nx_opd*nv_opd] x3 (marker, total, perturbed on (x,v))
                        float64[nv_opd] x3        (same in v)

All numbers come from the engine (GPU); this module only formats bytes.
"""
import numpy as np

VEC_FILE_CLASSID = 1211214


def _be(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype).astype(np.dtype(dtype).newbyteorder(">")).tobytes()


class OutputWriter:
    """output_init + output_field + output_ptcldist for one run"""

    def __init__(self, path, inp):
        self.inp = inp
        self.f = open(path, "wb")
        nm = inp.nmode
        ints = [inp.nspecies, nm, inp.nx, inp.nv, inp.nx_opd, inp.nv_opd] + [inp.modes[m] for m in range(nm)]
        self.f.write(_be(ints, np.int32))
        self.f.write(_be([inp.lx, inp.v_max], np.float64))
        self.nrecords = 0

    def _vec(self, a):
        a = np.asarray(a, dtype=np.float64)
        self.f.write(_be([VEC_FILE_CLASSID, a.size], np.int32))
        self.f.write(_be(a, np.float64))

    def write_record(self, engine):
        """output_all minus the progress line: one record from the engine's
        current state.  Returns int E^2 dx (what output_progress prints)."""
        if hasattr(engine, "output_all"):             # one call, one wait (pic1dp_hip_output_all)
            scal, fld, dists = engine.output_all()
        else:                                         # (any object with the three calls of the record)
            scal, fld = engine.output_scalars(), engine.get_field()
            dists = [engine.ptcldist(s, finish=True) for s in range(self.inp.nspecies)]
        self.f.write(_be(scal, np.float64))
        self._vec(fld["mode_re"])
        self._vec(fld["mode_im"])
        self._vec(fld["electric"])
        self._vec(fld["chargeden"])
        for d in dists:
            for k in ("markr_xv", "total_xv", "pertb_xv", "markr_v", "total_v", "pertb_v"):
                self.f.write(_be(d[k], np.float64))
        self.nrecords += 1
        return float(scal[1])

    def close(self):
        if self.f:
            self.f.close()
            self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def record_bytes(inp):
    nm, nx, ns = inp.nmode, inp.nx, inp.nspecies
    nxv, nv = inp.nx_opd * inp.nv_opd, inp.nv_opd
    return 8 * (2 + 3 * ns) + 2 * (8 + 8 * nm) + 2 * (8 + 8 * nx) + ns * 8 * (3 * nxv + 3 * nv)


def header_bytes(inp):
    return 4 * (6 + inp.nmode) + 16


class OutputData:
    """reader with the layout of tools/OutputData.py:26-79 (for tests and for
    users without the reference's Python 2 tools)"""

    def __init__(self, path):
        raw = open(path, "rb").read()
        pos = 0

        def take(dtype, n):
            nonlocal pos
            dt = np.dtype(dtype).newbyteorder(">")
            a = np.frombuffer(raw, dtype=dt, count=n, offset=pos).astype(dtype)
            pos += n * dt.itemsize
            return a

        head = take(np.int32, 6)
        self.nspecies, self.nmode, self.nx, self.nv, self.nx_opd, self.nv_opd = (int(v) for v in head)
        self.modes = take(np.int32, self.nmode)
        self.lx, self.v_max = (float(v) for v in take(np.float64, 2))
        ns, nxv = self.nspecies, self.nx_opd * self.nv_opd
        self.scalars, self.mode_re, self.mode_im, self.electric, self.chargeden = [], [], [], [], []
        self.ptcldist = []

        def vec(n_expected):
            cid, n = take(np.int32, 2)
            if cid != VEC_FILE_CLASSID or n != n_expected:
                raise ValueError("bad Vec header (%d, %d) at byte %d" % (cid, n, pos))
            return take(np.float64, int(n))

        while pos < len(raw):
            self.scalars.append(take(np.float64, 2 + 3 * ns))
            self.mode_re.append(vec(self.nmode))
            self.mode_im.append(vec(self.nmode))
            self.electric.append(vec(self.nx))
            self.chargeden.append(vec(self.nx))
            rec = []
            for _ in range(ns):
                d = {}
                for k in ("markr_xv", "total_xv", "pertb_xv"):
                    d[k] = take(np.float64, nxv).reshape(self.nv_opd, self.nx_opd)
                for k in ("markr_v", "total_v", "pertb_v"):
                    d[k] = take(np.float64, self.nv_opd)
                rec.append(d)
            self.ptcldist.append(rec)
        self.scalars = np.array(self.scalars)
        self.ntime = len(self.scalars)

    def get_scalar_t(self):
        """rows: time, int E^2 dx, then the species sums (tools/OutputData.py)"""
        return self.scalars.T

    def growthrate_energy_fit(self, time1, time2):
        """least-squares slope of ln(int E^2 dx), tools/OutputData.py:153-170"""
        t, e = self.scalars[:, 0], self.scalars[:, 1]
        i1 = int(np.searchsorted(t, time1)) - 1
        i2 = int(np.searchsorted(t, time2))
        tt, ln = t[i1:i2], np.log(e[i1:i2])
        n = i2 - i1
        return (n * np.sum(tt * ln) - np.sum(tt) * np.sum(ln)) / (n * np.sum(tt * tt) - np.sum(tt) ** 2)


def _es12_3e3(x):
    """Fortran edit descriptor es12.3e3"""
    if x == 0.0:
        return "%12s" % "0.000E+000"
    m, e = ("%.3e" % x).split("e")
    return "%12s" % ("%sE%s%03d" % (m, "+" if int(e) >= 0 else "-", abs(int(e))))


def progress_header():
    """output_progress(0), src/pic1dp_output.F90:521"""
    return "Info: progress:\nprogrss  itime     time  int E^2 dx\n"


def progress_line(inp, itime, time, electric_energy):
    """output_progress(1) at verbosity 1: format '(a, f5.1, a, i7, f9.3, es12.3e3, a)'
    (src/pic1dp_output.F90:510-526)"""
    pi = 1e2 * float(itime) / inp.ntime_max
    pt = 1e2 * time / inp.time_max
    c, pct = ("i", pi) if pi >= pt else ("t", pt)
    return "%s%5.1f%%%7d%9.3f%s\n" % (c, pct, itime, time, _es12_3e3(electric_energy))


def progress_line_optimized(inp, itime, time, nparticle_allspec):
    """output_progress(2) at verbosity 1, printed inside the step that performed a
    merge / remove / split: '(a, f5.1, a, i7, f9.3, a, i9, a)' with itime + 1 and
    time + dt (src/pic1dp_output.F90:527-532)"""
    pi = 1e2 * float(itime) / inp.ntime_max
    pt = 1e2 * time / inp.time_max
    c, pct = ("i", pi) if pi >= pt else ("t", pt)
    return "%s%5.1f%%%7d%9.3f : optimization performed, current # of particles %9d\n" % (
        c, pct, itime + 1, time + inp.dt, nparticle_allspec)
