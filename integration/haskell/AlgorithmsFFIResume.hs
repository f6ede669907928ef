-- NOT COMPILED IN THIS REPOSITORY'S IMAGE (no GHC here): this is the binding a maintainer of
-- jinilover/floydWarshall would add, kept as a file so that it can be dropped into src/lib/.
-- See INTEGRATION.md section 2c.  cabal: extra-libraries: fwx, build-depends: vector.
{-# LANGUAGE ForeignFunctionInterface #-}
-- Resuming variant (INTEGRATION.md section 2c).  The reference re-runs `floydWarshall` on the whole
-- rebuilt matrix after every accepted price update (ProcessRequests.hs:82-84 after :99-102).  When the
-- update is between two KNOWN vertices only two entries of `buildMatrix`'s output change
-- (Algorithms.hs:36-37), and they are operands of steps i and j only (Algorithms.hs:58-60): a host
-- that keeps one solver across updates sends the two entries and lets the engine resume at the last
-- checkpoint the change cannot have influenced.  Same bits as a fresh `floydWarshall` of the changed
-- map (tests/test_gpu_resume.py), `_path` lists included.
--
-- Use from ProcessRequests.syncMatrix (OutSync exRates):
--   * the vertex set is unchanged and exactly the pair (src,dest) / (dest,src) differs from the map
--     the solver was built from  ->  resolveEdge solver i j fwdR bkdR
--   * otherwise (a new vertex renumbers the matrix)  ->  newSolver (buildMatrix ...)
-- and read entries with `entryOf` where `optimum` forces them (as AlgorithmsFFILazy does).
module AlgorithmsFFIResume (Solver, newSolver, resolveEdge, entryOf, solverOrder) where

import           Control.Monad                (when)
import           Data.Int                     (Int32, Int64)
import qualified Data.Vector                  as V
import qualified Data.Vector.Storable         as S
import           Foreign.C.Types              (CInt (..))
import           Foreign.Concurrent           (newForeignPtr)
import           Foreign.ForeignPtr           (ForeignPtr, withForeignPtr)
import           Foreign.Marshal.Alloc        (alloca)
import           Foreign.Marshal.Array        (allocaArray, peekArray, withArray)
import           Foreign.Ptr                  (Ptr, nullPtr)
import           Foreign.Storable             (peek)

import           Types                        (Matrix, RateEntry (..), Vertex)

data FwxMatrix                                   -- opaque fwx_matrix
foreign import ccall safe "fwx.h fwx_matrix_create"
  c_create  :: Ptr (Ptr FwxMatrix) -> Int32 -> Int32 -> Int32 -> Int32 -> Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_keep_input"      c_keep_input    :: Ptr FwxMatrix -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_enable_path_log" c_enable_log    :: Ptr FwxMatrix -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_enable_resume"   c_enable_resume :: Ptr FwxMatrix -> Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_upload"
  c_upload  :: Ptr FwxMatrix -> Ptr Double -> Ptr Int32 -> Ptr Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_solve"   c_solve :: Ptr FwxMatrix -> Ptr () -> IO CInt
-- int fwx_matrix_resolve(fwx_matrix*, int32_t count, const int64_t* index, const void* rate_vals,
--                        const int32_t* next_vals, const int32_t* hops_vals, const fwx_opts*,
--                        int32_t* resumed_from);
foreign import ccall safe "fwx.h fwx_matrix_resolve"
  c_resolve :: Ptr FwxMatrix -> Int32 -> Ptr Int64 -> Ptr Double -> Ptr Int32 -> Ptr Int32
            -> Ptr () -> Ptr Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_query_exact"
  c_query   :: Ptr FwxMatrix -> Int32 -> Int32 -> Ptr Double -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "fwx.h fwx_matrix_destroy" c_destroy :: Ptr FwxMatrix -> IO CInt

fwxErrCapacity, fwxErrUnsupported :: CInt
fwxErrCapacity    = -6                            -- FWX_ERR_CAPACITY (fwx.h)
fwxErrUnsupported = -7                            -- FWX_ERR_UNSUPPORTED

-- | One solved rate matrix kept on the device, with the vertices in `buildMatrix`'s order.
data Solver = Solver { _handle :: ForeignPtr FwxMatrix, _vertices :: V.Vector Vertex }

solverOrder :: Solver -> V.Vector Vertex
solverOrder = _vertices

ok :: String -> CInt -> IO ()
ok what rc = when (rc /= 0) $ ioError (userError ("libfwx " ++ what ++ ": status " ++ show rc))

-- | `floydWarshall` of `buildMatrix`'s output, solved once, ready for `resolveEdge`.
-- (An empty matrix needs no solver: the caller keeps answering from the empty map.)
newSolver :: Matrix RateEntry -> IO Solver
newSolver m = do
  let n        = V.length m
      vertices = V.map (_start . V.head) m
      vIdx v   = maybe (-1) fromIntegral (V.elemIndex v vertices) :: Int32
      headIdx e = case _path e of { [] -> -1; (v:_) -> vIdx v }
      rate = S.fromList [ _bestRate e | row <- V.toList m, e <- V.toList row ]
      next = S.fromList [ headIdx e   | row <- V.toList m, e <- V.toList row ] :: S.Vector Int32
  h  <- alloca $ \pp -> do { ok "create" =<< c_create pp (fromIntegral n) 1 1 0 (-1); peek pp }  -- f64, next-hops
  fp <- newForeignPtr h (c_destroy h >> return ())
  ok "keep_input" =<< c_keep_input h                 -- the uploaded input stays on the device
  ok "enable_path_log" =<< c_enable_log h            -- exact `_path` lists (must precede enable_resume)
  -- 7 checkpoints + the panels of every pivot.  Returns the number of checkpoints (>= 0), or a negative
  -- status: FWX_ERR_UNSUPPORTED for n <= 64 (solved in one launch; any larger n works, odd ones included: the
  -- handle pads its rows), FWX_ERR_OOM when the device cannot hold them.  Resuming is an optimisation: whatever
  -- the status, the solver works, every update just re-solves from pivot 0 (fwx.h: fwx_matrix_resume_bytes /
  -- fwx_device_memory size the count to the device beforehand).
  _ <- c_enable_resume h 7
  S.unsafeWith rate $ \pr -> S.unsafeWith next $ \pn -> ok "upload" =<< c_upload h pr pn nullPtr
  ok "solve" =<< c_solve h nullPtr
  return (Solver fp vertices)

-- | The price of the pair (i, j) changed to `fwd` (i -> j) and `bkd` (j -> i): Algorithms.hs:36-37
-- puts (rate, [vtxJ]) at (i, j) and (rate, [vtxI]) at (j, i).  Re-solves; returns the pivot the
-- solve resumed at (0 = from scratch).  Every entry read afterwards is the new solution's.
resolveEdge :: Solver -> Int -> Int -> Double -> Double -> IO Int
resolveEdge (Solver fp vertices) i j fwd bkd =
  withForeignPtr fp $ \h ->
    withArray [fromIntegral (i * n + j), fromIntegral (j * n + i)] $ \pidx ->
    withArray [fwd, bkd] $ \prate ->
    withArray [fromIntegral j, fromIntegral i] $ \pnext ->
    alloca $ \pfrom -> do
      ok "resolve" =<< c_resolve h 2 pidx prate pnext nullPtr nullPtr pfrom
      fromIntegral <$> peek pfrom
  where n = V.length vertices

-- | Entry (i, j) of the solved matrix, `_path` exactly as the reference's list concatenation
-- (Algorithms.hs:55) builds it.  A negative status is an ERROR, never an empty path.
entryOf :: Solver -> Int -> Int -> IO RateEntry
entryOf (Solver fp vertices) i j = withForeignPtr fp $ \h -> go h (max 64 (4 * V.length vertices))
  where
    go h cap =
      alloca $ \pr -> allocaArray cap $ \pp -> do
        len <- c_query h (fromIntegral i) (fromIntegral j) pr pp (fromIntegral cap)
        if len == fwxErrCapacity && cap < 16777216 then go h (cap * 8)
        else if len < 0 then ioError (userError ("libfwx query_exact: status " ++ show len))
        else do
          r  <- peek pr
          ix <- peekArray (fromIntegral len) pp
          return (RateEntry r (vertices V.! i) [ vertices V.! fromIntegral x | x <- ix ])
