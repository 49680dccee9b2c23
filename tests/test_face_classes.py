"""Pack-time velocity-step classes of interior faces (radiative3d_amd/csrc/r3d_pack.h
classify_velocity_step): SMOOTH / STEP may only be set where the reference's run-time test
(phonons.cpp:243-252, max over P,S of |2 (v2-v1)/(v2+v1)| > 1e-5) has that outcome everywhere
on the face; everything else keeps the run-time test."""
import emul_ffi as E
from radiative3d_amd import _ffi


def test_corner_rules():
    big, small = 3e-5, 1e-6
    # all corners small for both types -> smooth
    assert E.class_from_corners([(small, -small), (small, small), (-small, small)]) == E.F_SMOOTH
    # one type beyond the threshold with one sign everywhere -> step, whatever the other does
    assert E.class_from_corners([(big, small), (2 * big, -small), (big, 0.0)]) == E.F_STEP
    assert E.class_from_corners([(small, -big), (-small, -big), (0.0, -2 * big)]) == E.F_STEP
    # |step| large at every corner but the sign changes across the face: it passes through
    # zero in between, so neither class holds (the reference hands over plainly there)
    assert E.class_from_corners([(big, -big), (-big, big), (big, -big)]) == 0
    assert E.class_from_corners([(big, big), (-big, -big), (big, big)]) == 0
    # straddling the threshold, or inside the guard band
    assert E.class_from_corners([(small, small), (big, small), (small, small)]) == 0
    assert E.class_from_corners([(1.05e-5, 0.0)]) == 0
    assert E.class_from_corners([(0.95e-5, 0.0)]) == 0
    assert E.class_from_corners([(float("nan"), 0.0)]) == 0


def test_builtin_models_keep_their_classes(models):
    """Layered models: every undisturbed layer boundary is a STEP (Appendix A item 4); tetra
    crust-pinch: faces inside a continuous gradient are SMOOTH; faces flagged as grid
    discontinuities or free surface carry no class."""
    m = models("lopnor")
    n_step = 0
    for ci in range(m.n_cells):
        c = m.desc.cells[ci]
        for f in range(2):
            cls = E.face_class(m, ci, f)
            fl = c.faces[f].flags
            if not (fl & _ffi.R3D_FACE_ADJOIN) or fl & (_ffi.R3D_FACE_DISCON | _ffi.R3D_FACE_REFLECT):
                assert cls == 0
            else:
                n_step += cls == E.F_STEP
    assert n_step > 0
    m = models("crustpinch")
    classes = [E.face_class(m, ci, f) for ci in range(m.n_cells) for f in range(4)]
    assert classes.count(E.F_SMOOTH) > classes.count(0) > 0
