"""pic1dp.out byte layout (src/pic1dp_output.F90:74-92,173-186,457-474; reader
tools/OutputData.py:26-79) and the progress line format -- CPU only, with a
stand-in engine that returns fixed numbers."""
import struct

import numpy as np


class FakeEngine:
    def __init__(self, inp):
        self.inp = inp
        self.k = 0

    def output_scalars(self):
        self.k += 1
        return np.arange(2 + 3 * self.inp.nspecies, dtype=np.float64) + 0.5 * self.k

    def get_field(self):
        nx, nm = self.inp.nx, self.inp.nmode
        return dict(mode_re=np.arange(nm) + 0.25, mode_im=-np.arange(nm) - 0.75,
                    electric=np.sin(np.arange(nx)), chargeden=np.cos(np.arange(nx)))

    def ptcldist(self, s, finish=True):
        nxv, nv = self.inp.nx_opd * self.inp.nv_opd, self.inp.nv_opd
        return dict(markr_xv=np.arange(nxv) + s, total_xv=np.arange(nxv) * 2.0, pertb_xv=-np.arange(nxv) * 1.0,
                    markr_v=np.arange(nv) + 10.0, total_v=np.arange(nv) + 20.0, pertb_v=np.arange(nv) + 30.0)


def test_record_size_matches_reference_default(amd):
    from pic1dp_amd import output
    inp = amd.make_input()
    assert output.record_bytes(inp) == 103000          # one default record
    assert output.header_bytes(inp) == 4 * 7 + 16


def test_writer_bytes_and_reader_roundtrip(amd, tmp_path):
    from pic1dp_amd import output
    inp = amd.make_input(nx=24, nmode=2, modes=[1, 3], nspecies=2, nx_opd=8, nv_opd=6,
                         species_charge=[-1.0, 1.0], species_mass=[1.0, 9.0], species_density=[0.9, 0.9],
                         species_temperature=[1.0, 1.0], species_temperature2=[1.0, 1.0], species_v0=[5.0, 0.0])
    path = tmp_path / "pic1dp.out"
    eng = FakeEngine(inp)
    with output.OutputWriter(str(path), inp) as w:
        e1 = w.write_record(eng)
        w.write_record(eng)
    raw = path.read_bytes()
    assert len(raw) == output.header_bytes(inp) + 2 * output.record_bytes(inp)
    # header: big-endian int32 x (6 + nmode), float64 x 2
    assert struct.unpack(">8i", raw[:32]) == (2, 2, 24, 128, 8, 6, 1, 3)
    assert struct.unpack(">2d", raw[32:48]) == (inp.lx, 8.0)
    # first record: scalars then Vec(mode_re) with PETSc's VEC_FILE_CLASSID
    assert struct.unpack(">8d", raw[48:112]) == tuple(np.arange(8) + 0.5)
    assert struct.unpack(">2i", raw[112:120]) == (1211214, 2)
    assert e1 == 1.5
    d = output.OutputData(str(path))
    assert (d.nspecies, d.nmode, d.nx, d.nv, d.nx_opd, d.nv_opd) == (2, 2, 24, 128, 8, 6)
    assert list(d.modes) == [1, 3] and d.ntime == 2
    assert np.array_equal(d.scalars[1], np.arange(8) + 1.0)
    assert np.array_equal(d.electric[0], np.sin(np.arange(24)))
    assert np.array_equal(d.mode_im[1], [-0.75, -1.75])
    assert d.ptcldist[0][1]["markr_xv"].shape == (6, 8)
    assert d.ptcldist[0][1]["markr_xv"][2, 3] == 2 * 8 + 3 + 1      # index iv*nx_opd + ix, species 1
    assert np.array_equal(d.ptcldist[1][0]["pertb_v"], np.arange(6) + 30.0)
    assert d.get_scalar_t().shape == (8, 2)


def test_progress_line_format(amd):
    """'(a, f5.1, a, i7, f9.3, es12.3e3, a)', src/pic1dp_output.F90:523-525"""
    from pic1dp_amd import output
    inp = amd.make_input()
    assert output.progress_line(inp, 10, 0.5, 6.44694586e-09) == "t  0.1%     10    0.500  6.447E-009\n"
    assert output.progress_line(inp, 0, 0.0, 6.7730612208e-09) == "i  0.0%      0    0.000  6.773E-009\n"
    assert output.progress_line(inp, 10000, 500.0, 1.2345e+02) == "t100.0%  10000  500.000  1.234E+002\n" \
        or output.progress_line(inp, 10000, 500.0, 1.2345e+02) == "t100.0%  10000  500.000  1.235E+002\n"
    assert output.progress_header().startswith("Info: progress:\nprogrss  itime")


def test_progress_line_optimized_format(amd):
    from pic1dp_amd import output
    inp = amd.make_input(nparticle_max=1000)
    assert output.progress_line_optimized(inp, 99, 4.95, 123456) == \
        "t  1.0%    100    5.000 : optimization performed, current # of particles    123456\n"
