"""Harness that runs the *reference* Nano-kappa code (read-only at /root/reference)
in THIS container only, to produce golden vectors for tests/golden/*.npz.

Must be executed with the interpreter that can run the reference
(`/opt/conda/bin/python3.9`: numpy 1.26, scipy 1.7 -- the reference needs
NumPy < 2).  Nothing here is imported by the package, the
tests or the bench: the reference never travels to the GPU box.

What is patched (harness-side only, no reference file is modified or copied):
  * `trimesh`, `shapely`, `phonopy` are absent here -> stub modules (they are only
    touched for STL loading, a discarded sanity check, and the HDF5/POSCAR loader);
  * plotting methods are replaced by no-ops (matplotlib 3.4.3 rejects `layout=`);
  * `Phonon` is built with `Phonon.__new__` and filled from the synthetic material
    (the Si/Ge HDF5 blobs are missing from the reference checkout), then the
    reference's own table builders are called.
"""
import os
import sys
import types
import tempfile

import numpy as np

REF = '/root/reference'
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))


def _install_stubs():
    if 'trimesh' not in sys.modules:
        sys.modules['trimesh'] = types.ModuleType('trimesh')
    if 'shapely' not in sys.modules:
        sh = types.ModuleType('shapely')
        shg = types.ModuleType('shapely.geometry')

        class Polygon(object):
            def __init__(self, *a, **k):
                pass

            def simplify(self, *a, **k):
                return self

            def equals(self, other):
                return True
        shg.Polygon = Polygon
        sh.geometry = shg
        sys.modules['shapely'] = sh
        sys.modules['shapely.geometry'] = shg
    if 'phonopy' not in sys.modules:
        ph = types.ModuleType('phonopy')
        ph.Phonopy = object
        phi = types.ModuleType('phonopy.interface')
        phc = types.ModuleType('phonopy.interface.calculator')
        phc.read_crystal_structure = lambda *a, **k: None
        ph.interface = phi
        phi.calculator = phc
        sys.modules['phonopy'] = ph
        sys.modules['phonopy.interface'] = phi
        sys.modules['phonopy.interface.calculator'] = phc


def import_reference():
    _install_stubs()
    import matplotlib
    matplotlib.use('Agg')
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if REPO not in sys.path:
        sys.path.insert(1, REPO)
    import warnings
    warnings.simplefilter('ignore')
    from classes.Geometry import Geometry
    from classes.Phonon import Phonon
    from classes.Population import Population
    from classes.Visualisation import Visualisation
    import argument_parser

    def _noop(*a, **k):
        return None

    Geometry.plot_mesh_bc = _noop
    Geometry.save_subvol_connections = _noop
    Population.plot_figures = _noop
    for name in ('plot_convergence_general', 'convergence_energy_balance',
                 'flux_contribution', 'plot_kappa_path'):
        setattr(Visualisation, name, _noop)

    # find_specular_correspondences draws a debug figure per normal; keep the maths,
    # drop the figure by handing it a do-nothing pyplot facade.
    import classes.Population as popmod

    class _Ax(object):
        def __getattr__(self, name):
            return _noop

    class _AxGrid(object):
        def __getitem__(self, idx):
            return _Ax()

        def ravel(self):
            return []

    class _Plt(object):
        def subplots(self, *a, **k):
            return _Ax(), _AxGrid()

        def __getattr__(self, name):
            return _noop
    popmod.plt = _Plt()
    return types.SimpleNamespace(Geometry=Geometry, Phonon=Phonon, Population=Population,
                                 Visualisation=Visualisation, argument_parser=argument_parser)


def make_args(ref, argv, results_folder=None):
    parser = ref.argument_parser.initialise_parser(False)
    args = parser.parse_args(argv)
    if results_folder is None:
        results_folder = tempfile.mkdtemp(prefix='nkref_')
    args.results_folder = results_folder
    return args


def make_phonon(ref, args, material):
    """Fill a reference Phonon object from FBZ-expanded tables and run the reference's
    own builders (Phonon.py:115-149)."""
    Phonon = ref.Phonon
    p = Phonon.__new__(Phonon)
    super(Phonon, p).__init__()      # Constants
    p.args = args
    p.mat_index = 0
    p.mat_folder = args.results_folder
    p.data_mesh = np.array(material['data_mesh'])
    p.q_points = np.array(material['q_points'], dtype=float)
    p.weights = np.ones(p.q_points.shape[0])
    p.frequency = np.array(material['frequency'], dtype=float)
    p.convert_to_omega()
    # keep omega bit-identical to the tables handed to the build
    p.omega = np.array(material['omega'], dtype=float)
    p.group_vel = np.array(material['group_vel'], dtype=float)
    p.temperature_array = np.array(material['temperature'], dtype=float)
    p.gamma = np.array(material['gamma'], dtype=float)
    p.volume_unitcell = float(material['volume_unitcell'])
    p.number_of_qpoints = p.q_points.shape[0]
    p.number_of_branches = p.frequency.shape[1]
    p.number_of_modes = p.number_of_qpoints * p.number_of_branches
    p.inactive_modes_mask = np.all(p.group_vel == 0, axis=2)
    p.number_of_inactive_modes = p.inactive_modes_mask.sum()
    p.number_of_active_modes = p.number_of_modes - p.number_of_inactive_modes
    p.reciprocal_lattice = np.array(material['reciprocal_lattice'], dtype=float)
    p.unique_modes = np.stack(np.meshgrid(np.arange(p.number_of_qpoints),
                                          np.arange(p.number_of_branches)), axis=-1).reshape(-1, 2).astype(int)
    p.get_wavevectors()
    p.get_norms()
    p.find_degeneracies()
    p.calculate_lifetime()
    p.zero_point = p.calculate_zeropoint()
    p.initialise_temperature_function()
    p.initialise_density_of_states()
    return p


from ref_harness_args import BOX_TTP, BOX_TTRRP, COMMON, argv_for  # noqa: E402,F401
