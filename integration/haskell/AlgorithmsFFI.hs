-- NOT COMPILED IN THIS REPOSITORY'S IMAGE (no GHC here): this is the binding a maintainer of
-- jinilover/floydWarshall would add, kept as a file so that it can be dropped into src/lib/.
-- See INTEGRATION.md.  cabal: extra-libraries: fwx, build-depends: vector.
{-# LANGUAGE ForeignFunctionInterface #-}
-- src/lib/AlgorithmsFFI.hs  (new file; Algorithms.floydWarshall then becomes
--   floydWarshall = runAlgoGPU . buildMatrix)
module AlgorithmsFFI (runAlgoGPU) where

import           Data.Int                     (Int32)
import qualified Data.Vector                  as V
import qualified Data.Vector.Storable         as S
import qualified Data.Vector.Storable.Mutable as SM
import           Foreign.C.Types              (CInt (..))
import           Foreign.Ptr                  (Ptr, nullPtr)
import           System.IO.Unsafe             (unsafePerformIO)

import           Types                        (Matrix, RateEntry (..), Vertex)

-- int fwx_solve_f64(int32_t n, double*, int32_t* next, int32_t* hops, const fwx_opts*);
foreign import ccall safe "fwx.h fwx_solve_f64"
  c_fwx_solve_f64 :: Int32 -> Ptr Double -> Ptr Int32 -> Ptr Int32 -> Ptr () -> IO CInt

-- | Drop-in for `runAlgo 0` (Algorithms.hs:42-61) on the output of buildMatrix (:26-40).
runAlgoGPU :: Matrix RateEntry -> Matrix RateEntry
runAlgoGPU m
  | n == 0    = m                                     -- AlgorithmsTest.hs:62-64
  | otherwise = unsafePerformIO $ do
      rate <- S.thaw (S.fromList [ _bestRate e | row <- V.toList m, e <- V.toList row ])
      next <- S.thaw (S.fromList [ headIdx e   | row <- V.toList m, e <- V.toList row ])
      rc <- SM.unsafeWith rate $ \pr -> SM.unsafeWith next $ \pn ->
              c_fwx_solve_f64 (fromIntegral n) pr pn nullPtr nullPtr
      if rc /= 0 then error ("fwx_solve_f64 failed: " ++ show rc) else do
        r  <- S.freeze rate
        nx <- S.freeze next
        return (V.generate n (\i -> V.generate n (\j -> entry r nx i j)))
  where
    n        = V.length m
    vertices = V.map (_start . V.head) m              -- row i starts at vertex i
    vIdx v   = maybe (-1) fromIntegral (V.elemIndex v vertices)
    headIdx e = case _path e of { [] -> -1; (v:_) -> vIdx v }
    -- `_path` = follow head-of-path from i until j  (Algorithms.hs:55 builds it by ++).  The walk
    -- is BOUNDED by n hops, like fwx_follow_path (FWX_ERR_CYCLE): an arbitrage cycle, which the
    -- parser admits across exchanges, would otherwise build an infinite list where the
    -- reference's `_path` is finite.  For the reference's exact lists under ties and cycles use
    -- AlgorithmsFFILazy (fwx_matrix_query_exact).
    entry r nx i j =
      let walk :: Int -> Int -> [Vertex]
          walk hopsLeft cur
            | nx S.! (cur*n + j) < 0 = []
            | hopsLeft == 0 = error "fwx: next-hop walk does not reach the destination (cycle)"
            | otherwise = let h = fromIntegral (nx S.! (cur*n + j))
                          in (vertices V.! h) : (if h == j then [] else walk (hopsLeft - 1) h)
      in RateEntry (r S.! (i*n + j)) (vertices V.! i) (if i == j then [] else walk n i)
