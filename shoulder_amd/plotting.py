"""`Plot` -- 3-D figure of a bone with its landmarks, or of a resected humerus.

Mirror of reference `src/shoulder/plotting.py:13-128` (same class names, arguments, trace settings and error text).  A
plotly figure is a host-side view of results the device path already produced; nothing here computes geometry.
plotly is imported on use and its absence is an ImportError, not a silent no-op.
"""
import numpy as np

from . import arthroplasty, base


def mesh2plotly(mesh):
    """plotting.py:13-25 (`trimesh2plotly`): a Mesh3d trace from vertices and faces."""
    import plotly.graph_objects as go
    v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
    return go.Mesh3d(x=v[:, 0], y=v[:, 1], z=v[:, 2], i=f[:, 0], j=f[:, 1], k=f[:, 2])


trimesh2plotly = mesh2plotly      # the reference's name


def bone_mesh_settings(trace):
    """plotting.py:28-42."""
    trace.color = "#DFDAC0"
    trace.lighting = dict(ambient=0.18, diffuse=0.8, fresnel=0.1, specular=0.6, roughness=0.05,
                          facenormalsepsilon=1e-15, vertexnormalsepsilon=1e-15)
    trace.lightposition = dict(x=1000, y=1000, z=-1000)
    trace.flatshading = False
    return trace


class PlotSurgery:
    """plotting.py:75-101: the two halves of `HumeralHeadOsteotomy.resect_mesh()`."""

    def __init__(self, ost, opacity):
        self.mesh_top, self.mesh_bot = ost.resect_mesh()
        self.opacity = opacity
        self.name = ost._humerus.stl_file.name

    @property
    def figure(self):
        import plotly.graph_objects as go
        fig = go.Figure()
        top = bone_mesh_settings(mesh2plotly(self.mesh_top))
        top.opacity = self.opacity
        bot = bone_mesh_settings(mesh2plotly(self.mesh_bot))
        fig.add_traces([top, bot])
        return fig


class PlotLandmarks:
    """plotting.py:104-128: the bone plus the graph object(s) of every landmark computed so far."""

    def __init__(self, bone, opacity):
        self.mesh = bone.mesh
        self.opacity = opacity
        self.name = bone.stl_file.name
        self._landmarks_graph_obj = bone._list_landmarks_graph_obj()

    @property
    def figure(self):
        import plotly.graph_objects as go
        fig = go.Figure()
        m = bone_mesh_settings(mesh2plotly(self.mesh))
        m.opacity = self.opacity
        fig.add_trace(m)
        for lgo in self._landmarks_graph_obj:
            for tr in (lgo if isinstance(lgo, list) else [lgo]):
                fig.add_trace(tr)
        return fig


class Plot:
    """plotting.py:45-72.  obj2plot: a Bone or a HumeralHeadOsteotomy; opacity of the bone (default 0.7)."""

    def __init__(self, obj2plot, opacity=0.7):
        if isinstance(obj2plot, arthroplasty.HumeralHeadOsteotomy):
            self._plotter = PlotSurgery(obj2plot, opacity)
        elif isinstance(obj2plot, base.Bone):
            self._plotter = PlotLandmarks(obj2plot, opacity)
        else:
            raise ValueError("Object to plot must be either a Bone or HumeralHeadOjson")      # (the reference's text)
        self.figure = self._plotter.figure
        self.figure.update_layout(title=self._plotter.name, scene_aspectmode="data")
