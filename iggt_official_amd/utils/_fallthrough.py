"""Keep the drop-in import paths complete: an alias module that implements only the functions on the hot path fills in
every OTHER public name from the reference's module of the same name when a reference checkout is importable behind this
repository (iggt.__path__ is extended over sys.path, iggt/__init__.py).  Without a checkout those names are simply absent.
Names this repository implements always win (tests/test_schema.py); what was taken from where is recorded in `IMPORTED` and
logged (logger "iggt_official_amd.fallthrough", level INFO); only an ImportError of the reference module (its optional
third-party packages) is tolerated -- any other failure propagates."""
import importlib.util
import logging
import os
import sys

log = logging.getLogger("iggt_official_amd.fallthrough")
IMPORTED = {}   # "<package>.<module>" -> (path of the reference file, [names taken from it]) or (path, "failed: <error>")


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
        except ImportError as ex:   # the reference module needs packages that are not installed (cv2, torch_geometric, ...)
            IMPORTED[f"{package}.{module}"] = (path, f"failed: {ex!r}")
            log.info("%s.%s: reference module %s not importable (%r); only this repository's names are available",
                     package, module, path, ex)
            return
        taken = []
        for k, v in vars(mod).items():
            if not k.startswith("_") and k not in namespace:
                namespace[k] = v
                taken.append(k)
        IMPORTED[f"{package}.{module}"] = (path, taken)
        log.info("%s.%s: %d name(s) filled in from the reference checkout %s: %s", package, module, len(taken), path,
                 ", ".join(sorted(taken)[:12]) + (" ..." if len(taken) > 12 else ""))
        return
