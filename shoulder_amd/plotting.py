"""`Plot` -- a plotly figure of a bone with the landmarks computed so far, or of the two halves of a resection.

Public surface of reference `src/shoulder/plotting.py:13-128`: `Plot(obj, opacity=0.7).figure`, `PlotLandmarks`,
`PlotSurgery`, `trimesh2plotly`, `bone_mesh_settings`, the same bone colour / lighting / trace names / layout and the
same ValueError for anything that is neither a Bone nor a HumeralHeadOsteotomy.  A figure is a host-side view of
results the device path already produced; nothing here computes geometry.  plotly is imported on first use and its
absence is an ImportError, not a silent no-op.
"""
import numpy as np

from . import arthroplasty, base

_BONE_COLOUR = "#DFDAC0"
_BONE_LIGHTING = dict(ambient=0.18, diffuse=0.8, fresnel=0.1, specular=0.6, roughness=0.05,
                      facenormalsepsilon=1e-15, vertexnormalsepsilon=1e-15)
_LIGHT_POSITION = dict(x=1000, y=1000, z=-1000)


def _go():
    import plotly.graph_objects as go
    return go


def trimesh2plotly(mesh):
    """vertices + faces -> `go.Mesh3d` (plotting.py:13-25); any object with `.vertices` (V,3) and `.faces` (F,3) will do"""
    xyz = np.asarray(mesh.vertices, dtype=np.float64).T
    ijk = np.asarray(mesh.faces).T
    return _go().Mesh3d(x=xyz[0], y=xyz[1], z=xyz[2], i=ijk[0], j=ijk[1], k=ijk[2])


mesh2plotly = trimesh2plotly


def bone_mesh_settings(trace):
    """bone colour, smooth shading and the reference's light (plotting.py:28-42)"""
    trace.update(color=_BONE_COLOUR, lighting=_BONE_LIGHTING, lightposition=_LIGHT_POSITION, flatshading=False)
    return trace


def _bone_trace(mesh, opacity=None):
    trace = bone_mesh_settings(trimesh2plotly(mesh))
    if opacity is not None:
        trace.opacity = opacity
    return trace


class PlotSurgery:
    """head (at the requested opacity) and resected humerus (opaque) of `ost.resect_mesh()` -- plotting.py:75-101"""

    def __init__(self, ost, opacity):
        self.mesh_top, self.mesh_bot = ost.resect_mesh()
        self.opacity = opacity
        self.name = ost._humerus.stl_file.name

    @property
    def figure(self):
        return _go().Figure(data=[_bone_trace(self.mesh_top, self.opacity), _bone_trace(self.mesh_bot)])


class PlotLandmarks:
    """the bone plus one trace (the anatomic neck: two) per landmark that has been computed -- plotting.py:104-128"""

    def __init__(self, bone, opacity):
        self.mesh = bone.mesh
        self.opacity = opacity
        self.name = bone.stl_file.name
        self._landmarks_graph_obj = bone._list_landmarks_graph_obj()

    @property
    def figure(self):
        traces = [_bone_trace(self.mesh, self.opacity)]
        for obj in self._landmarks_graph_obj:
            traces.extend(obj if isinstance(obj, list) else [obj])
        return _go().Figure(data=traces)


class Plot:
    """`Plot(bone_or_osteotomy, opacity=0.7).figure` -- plotting.py:45-72"""

    def __init__(self, obj2plot, opacity=0.7):
        if isinstance(obj2plot, arthroplasty.HumeralHeadOsteotomy):
            plotter = PlotSurgery(obj2plot, opacity)
        elif isinstance(obj2plot, base.Bone):
            plotter = PlotLandmarks(obj2plot, opacity)
        else:
            raise ValueError("Object to plot must be either a Bone or HumeralHeadOjson")      # the reference's text
        self._plotter = plotter
        self.figure = plotter.figure
        self.figure.update_layout(title=plotter.name, scene_aspectmode="data")      # "data": no distortion
