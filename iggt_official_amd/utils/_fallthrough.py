"""Keep the drop-in import paths complete: an alias module that implements only the functions on the hot path fills in
every OTHER public name from the reference's module of the same name when a reference checkout is importable behind this
repository (iggt.__path__ is extended over sys.path, iggt/__init__.py).  Without a checkout those names are simply absent."""
import importlib.util
import os
import sys


def fill_missing(package: str, module: str, namespace: dict) -> None:
    """Copy the public names of <other roots>/<package path>/<module>.py that `namespace` does not define."""
    pkg = sys.modules.get(package)
    roots = list(getattr(pkg, "__path__", []))[1:] if pkg is not None else []
    if not roots:
        top = package.split(".")[0]
        sub = os.path.join(*package.split(".")[1:]) if "." in package else ""
        here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        roots = [os.path.join(p, top, sub) for p in sys.path
                 if p and os.path.abspath(p) != here and os.path.isdir(os.path.join(p, top, sub))]
    for root in roots:
        path = os.path.join(root, module + ".py")
        if not os.path.exists(path):
            continue
        try:
            # loaded as a private sibling inside the package, so that the reference module's relative imports
            # (`from .rotation import ...`) resolve -- through the same extended package path
            name = f"{package}._reference_{module}"
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(name, None)
                raise
        except Exception:  # noqa: BLE001  (the reference module may need packages that are not installed)
            return
        for k, v in vars(mod).items():
            if not k.startswith("_") and k not in namespace:
                namespace[k] = v
        return
